"""The GPU parity suite runs EVERY fixture through all four kernel paths (tests/test_gpu_parity.py GOLDEN_CASES; rounds 4-5 kept part
of that matrix behind --runslow, round 6 folded it back).  This CPU test holds the matrix to that claim and checks that every
(fixture, path) pair lowers to a program -- or is refused by name -- without a GPU."""
from tests.test_gpu_parity import GOLDEN_CASES
from tests.util import FIXTURES, load_fixture


def test_the_parity_matrix_is_every_fixture_times_every_kernel_path():
    assert sorted(GOLDEN_CASES, key=str) == sorted(((n, f) for n in FIXTURES for f in (False, "trunk", True, "direct")), key=str)
    import pathlib
    src = "".join(p.read_text() for p in pathlib.Path(__file__).parent.glob("test_gpu_*.py"))
    assert "mark.slow" not in src and "mark.skip" not in src            # nothing of the GPU suite hides from the driver's run


def test_every_fixture_lowers_on_every_kernel_path():
    from hello_amd import compiler
    shapes = {}
    for name, fused in GOLDEN_CASES:
        spec, state, _, _ = load_fixture(name)
        prog = compiler.compile_model(spec, state, fused=True if fused == "direct" else fused, winograd=fused != "direct")
        assert prog.ops and prog.n_experts in (1, 3)
        shapes.setdefault(fused, set()).update((o.kind, o.cin, o.cout, o.k, o.stride, o.pad, o.lin, o.lout) for o in prog.ops)
    # the four paths really are different programs: the layer-by-layer one has no fused read convolver, the product path has one
    assert not any(k == compiler.OP_READCONV_FUSED for k, *_ in shapes[False]) and any(k == compiler.OP_READCONV_FUSED for k, *_ in shapes[True])
