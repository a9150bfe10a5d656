"""The host->device streaming pipeline returns exactly what a direct engine call returns, in order."""
import numpy as np
import pytest

from hello_amd import netspec as ns, synth, weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,kw", [
    ("single_tech", dict(coverage=30)),
    ("hybrid_ensemble2", dict(coverage=20, hybrid_coverage=10)),      # two read sets, one-hot reference, meta
])
def test_pipeline_matches_direct_calls_in_order(cfg, kw):
    import torch
    from hello_amd.engine import Engine
    from hello_amd.pipeline import HostPipeline
    spec = ns.build(cfg)
    state = weights.synth_state(spec, seed=5)
    eng = Engine(spec, state, device=0)
    batches = [synth.make_sites(n, seed=40 + i, **kw) for i, n in enumerate([64, 7, 200, 33, 1, 120])]
    direct = [eng.forward_batch(b, posteriors=True) for b in batches]
    pipe = HostPipeline(eng, depth=2)
    got = []
    for i, b in enumerate(batches):
        if i == 2:            # a producer that already writes pinned memory: no staging copy
            b = synth.SiteBatch(torch.from_numpy(b.reads0).pin_memory(), b.reads_per_allele0, b.alleles_per_site,
                                b.ref_onehot, b.reads1, b.reads_per_allele1)
        got += pipe.submit(b, tag=i)
    got += pipe.flush()
    assert [g[0] for g in got] == list(range(len(batches)))
    for (tag, logits, meta, post), (dl, dm, dp) in zip(got, direct):
        assert np.array_equal(logits, dl) and np.array_equal(post, dp)
        assert (meta is None and dm is None) or np.array_equal(meta, dm)
    assert pipe.flush() == []
    # several engines: consecutive batches run concurrently on their own compute streams, same answers, same order
    eng2 = Engine(spec, state, device=0)
    pipe = HostPipeline(engines=[eng, eng2])
    got = []
    for i, b in enumerate(batches * 2):
        got += pipe.submit(b, tag=i)
    got += pipe.flush()
    assert [g[0] for g in got] == list(range(2 * len(batches)))
    for (tag, logits, meta, post), (dl, dm, dp) in zip(got, direct * 2):
        assert np.array_equal(logits, dl) and np.array_equal(post, dp)
    eng2.close()
    eng.close()
