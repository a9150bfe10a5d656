"""the GPU parity suite runs its two intermediate kernel paths ("trunk": layered stem + fused trunk; "direct": fused,
direct-form convolutions) on a trimmed set of fixtures; the rest of the matrix sits behind --runslow.  This CPU test checks that
the trimming lost no kernel shape: it lowers every (fixture, path) pair of both sets and compares the op shapes."""
from tests.test_gpu_parity import GOLDEN_CASES, GOLDEN_CASES_REST
from tests.util import load_fixture


def test_every_kernel_shape_of_the_untrimmed_matrix_is_in_the_trimmed_one():
    """The set of op shapes (kind, channels, kernel, stride, lengths, flags) the two paths lower to on the
    dropped fixtures is a subset of what the default matrix lowers to."""
    from hello_amd import compiler

    def shapes(name, fused):
        spec, state, _, _ = load_fixture(name)
        try:
            prog = compiler.compile_model(spec, state, fused=True if fused == "direct" else fused, winograd=fused != "direct")
        except (ValueError, NotImplementedError):
            return set()
        return {(o.kind, o.cin, o.cout, o.k, o.stride, o.pad, o.lin, o.lout, o.flags & ~(64 | 128), o.c1) for o in prog.ops}
    covered = set()
    for n, f in GOLDEN_CASES:
        covered |= shapes(n, f)
    missing = {}
    for n, f in GOLDEN_CASES_REST:
        lost = shapes(n, f) - covered
        if lost:
            missing[(n, f)] = sorted(lost)
    assert not missing, missing
