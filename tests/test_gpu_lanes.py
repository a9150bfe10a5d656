"""Laned programs (small launches of multi-chain models): the independent chains of a two-technology / three-expert forward
(MixtureOfExpertsAdvanced.py:161-252) on their own streams, ordered by events where one reads another's output -- the same kernels,
the same bits as the sequential program, sooner."""
import numpy as np
import pytest

from hello_amd import compiler, netspec as ns, synth, weights
from tests.util import load_fixture

pytestmark = pytest.mark.gpu

MULTI_CHAIN = ["hybrid_no_ensemble", "hybrid_full", "hybrid_ensemble2", "hybrid_compressor2", "merged_hybrid", "hybrid_no_ensemble_wide",
               "merged_hybrid_250", "hybrid_no_ensemble_addendum"]


@pytest.mark.parametrize("name", MULTI_CHAIN)
def test_laned_program_gives_the_sequential_programs_bits(name):
    """Every multi-chain fixture through an engine that runs lanes (small launch: the default route) and through one that was handed
    the sequential program: logits, meta weights and pair posteriors equal bit for bit, on the fixture's batch and on fifty
    back-to-back launches of varying composition (a missing event would show as a changed bit sooner or later); and both within
    the golden tolerance of the reference's outputs."""
    from hello_amd.engine import Engine, LANES_MAX_SITES
    spec, state, batch, exp = load_fixture(name)
    laned = Engine(spec, state, device=0)
    sequential = Engine(spec, state, device=0, program=compiler.compile_model(spec, state))
    assert batch.n_sites <= LANES_MAX_SITES and sequential.small_launch_handle().value == sequential.handle.value
    got = laned.forward_batch(batch, posteriors=True)
    assert laned.lanes_handle is not None and laned.lanes_program.n_lanes >= 2 and laned._last_native.value == laned.lanes_handle.value
    want = sequential.forward_batch(batch, posteriors=True)
    for g, w in zip(got, want):
        assert (g is None) == (w is None) and (g is None or np.array_equal(g, w))
    np.testing.assert_allclose(got[0], exp["logits"], rtol=2e-5, atol=2e-4)
    kw = dict(coverage=20, hybrid_coverage=10, window=250) if name == "merged_hybrid_250" else dict(coverage=20, hybrid_coverage=10)
    pool = synth.make_sites(48, seed=5, **kw)
    rng = np.random.default_rng(1)
    for rep in range(50):
        lo = int(rng.integers(0, 40))
        sub = pool.site_slice(lo, lo + int(rng.integers(1, 9)))
        a, b = laned.forward_batch(sub, posteriors=True), sequential.forward_batch(sub, posteriors=True)
        assert all((x is None) == (y is None) and (x is None or np.array_equal(x, y)) for x, y in zip(a, b)), (name, rep)
    # per-op profiling and debug capture time / snapshot the sequential program: an engine with them armed leaves the lanes
    laned.set_profiling(4)
    assert laned.small_launch_handle().value == laned.handle.value
    laned.forward_batch(batch, posteriors=True)
    rows, n = laned.op_times_ms()
    assert n == 1 and len(rows) == len(laned.program.ops)
    laned.set_profiling(0)
    assert laned.small_launch_handle().value == laned.lanes_handle.value
    # a launch larger than the lanes' limit runs the sequential program (several engines sharing a card lose with lanes beyond the
    # latency regime: the default limit is 64 sites) -- and an engine that is told it has the card to itself keeps the lanes, with
    # the sequential program's bits
    mid = synth.make_sites(300, seed=6, **kw)
    b = sequential.forward_batch(mid, posteriors=True)
    laned.forward_batch(mid)
    assert laned._last_native.value == laned.handle.value
    laned.lanes_max_sites = 4096
    a = laned.forward_batch(mid, posteriors=True)
    assert laned._last_native.value == laned.lanes_handle.value
    assert all((x is None) == (y is None) and (x is None or np.array_equal(x, y)) for x, y in zip(a, b))
    laned.close()
    sequential.close()


def test_device_path_launches_on_lanes_stay_ordered_with_the_callers_stream():
    """Asynchronous device-path calls: outputs of back-to-back laned launches on the caller's stream are complete when that stream
    is (the call's stream joins every lane before the posteriors kernel), and equal the host path's."""
    import torch
    from hello_amd.engine import Engine
    spec = ns.build("hybrid_full")
    state = weights.synth_state(spec, seed=3)
    eng = Engine(spec, state, device=0)
    batches = [synth.make_sites(n, seed=20 + n, coverage=25, hybrid_coverage=12) for n in (3, 7, 1, 12, 5)]
    want = [eng.forward_batch(b, posteriors=True) for b in batches]
    stream = torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(stream):
        for b in batches * 4:
            dev = synth.SiteBatch(torch.from_numpy(b.reads0).cuda(), b.reads_per_allele0, b.alleles_per_site, torch.from_numpy(b.ref_onehot).cuda(),
                                  torch.from_numpy(b.reads1).cuda(), b.reads_per_allele1)
            outs.append(eng.forward_batch(dev, posteriors=True, stream=stream.cuda_stream))
    stream.synchronize()
    for k, got in enumerate(outs):
        for g, w in zip(got, want[k % len(batches)]):
            assert np.array_equal(g.cpu().numpy(), w)
    eng.close()
