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


def test_pipeline_sink_keeps_logits_resident():
    """The multi-GPU path's device sink: every batch's logits (and meta) also land in a caller-owned device tensor
    at the given allele column / site row, bit-identical to what the pipeline returns to the host."""
    import torch
    from hello_amd.engine import Engine
    from hello_amd.pipeline import HostPipeline, pin_batch
    spec = ns.build("hybrid_ensemble2")
    state = weights.synth_state(spec, seed=5)
    eng = Engine(spec, state, device=0)
    whole = pin_batch(synth.make_sites(90, seed=61, coverage=20, hybrid_coverage=10))
    cuts = [0, 40, 41, 90]
    pieces = [whole.site_slice(a, b) for a, b in zip(cuts, cuts[1:])]          # pinned views, no staging copy
    assert all(p.reads0.is_pinned() and p.reads0.is_contiguous() for p in pieces)
    sink_l = torch.zeros((eng.n_experts, whole.n_alleles), device="cuda")
    sink_m = torch.zeros((whole.n_sites, 3), device="cuda")
    pipe = HostPipeline(eng, depth=2)
    got, col, row = [], 0, 0
    for p in pieces:
        got += pipe.submit(p, sink=(sink_l, sink_m, col, row))
        col, row = col + p.n_alleles, row + p.n_sites
    got += pipe.flush()
    torch.cuda.synchronize()
    assert np.array_equal(sink_l.cpu().numpy(), np.concatenate([g[1] for g in got], axis=1))
    assert np.array_equal(sink_m.cpu().numpy(), np.concatenate([g[2] for g in got], axis=0))
    direct = eng.forward_batch(synth.SiteBatch(whole.reads0.numpy(), whole.reads_per_allele0, whole.alleles_per_site,
                                               whole.ref_onehot.numpy(), whole.reads1.numpy(), whole.reads_per_allele1))
    np.testing.assert_allclose(sink_l.cpu().numpy(), direct[0], rtol=1e-5, atol=1e-5)   # per-site sums re-associate
    eng.close()


def test_bench_under_torchrun_initialises_rccl_and_reports_one_gpu(tmp_path):
    """bench.py as the driver launches it for N > 1 (python -m torch.distributed.run, one process per GPU), here with
    one rank on this box's one GPU: a fresh child process (the launcher runs before anything touches the GPU), RCCL
    communicator up, the product's partitioner + pipeline + the one gather on the measured path, one JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), os.path.join(root, "bench.py"), "--gpus", "1",
           "--steps", "2", "--warmup", "1", "--sites", "512", "--launches-per-step", "3", "--no-secondary"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["sites_total"] == 2 * 3 * 512 and rec["config"]["repeat_passes_bit_identical"]
    assert rec["value"] > 0 and 0 < rec["roofline"]["frac"] <= 1.0 and rec["roofline"]["kernel"] == "readconv_kernel"


@pytest.mark.gpu
def test_bench_strong_scaling_mode_at_world_one_equals_the_plain_run(tmp_path):
    """``--scaling strong`` (the N = 1 stream cut N ways) under torch.distributed.run at world size 1
    is the plain run's workload -- the same number of sites, and its value within 3 % of the plain run's on the same box
    (so the N = 1 point of a scaling curve agrees with the headline bench)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-secondary", "--no-cpu-baseline"]
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, cwd=root, env=env, capture_output=True,
                           text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    strong = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                             "127.0.0.1", "--master-port", str(29900 + os.getpid() % 90), os.path.join(root, "bench.py")] + common +
                            ["--scaling", "strong"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert strong.returncode == 0, strong.stderr[-2000:]
    a = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in strong.stdout.splitlines() if ln.startswith("{")][-1])
    assert a["scaling"] == "weak" and b["scaling"] == "strong" and b["n_gpus"] == 1
    assert a["config"]["sites_total"] == b["config"]["sites_total"] == 6 * 10 * 8192
    assert abs(b["value"] / a["value"] - 1.0) < 0.03, (a["value"], b["value"])
    assert b["cpu_baseline"] is None and a["cpu_baseline"] is None          # switched off here; at N > 1 it carries a reason
