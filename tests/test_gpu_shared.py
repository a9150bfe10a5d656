"""hello_amd.shared on the GPU: ``loader.load(path, shared=True)`` -- the per-site call of the reference's worker processes
(caller_calling.py:863-868,872-891), scored by ONE server process per GPU that coalesces the workers' concurrent sites into one
launch -- against the direct per-site call of ``ScoringNetwork`` on the same sites."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from hello_amd import loader, netspec as ns, synth, weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _calls(config, n, seed, **kw):
    sys.path.insert(0, ROOT)
    import bench
    import torch
    out = []
    for fd, seg in bench.feature_dicts(synth.make_sites(n, seed=seed, **kw)):
        out.append(({a: (torch.from_numpy(f), None if g is None else torch.from_numpy(g)) for a, (f, g) in fd.items()}, torch.from_numpy(seg)))
    return out


def _as_arrays(result):
    mix, e0, e1, e2, meta = result
    return np.array([[float(v) for v in row.values()] for row in (mix, e0, e1, e2)], np.float32), meta.numpy().copy(), list(mix)


@pytest.fixture
def model_file(tmp_path, monkeypatch):
    monkeypatch.setenv("HELLO_SHARED_DIR", str(tmp_path / "rendezvous"))
    monkeypatch.setenv("HELLO_SHARED_IDLE_EXIT", "1")         # the box admits few processes on its card: a test's server leaves at once

    def make(config, seed=11):
        spec = ns.build(config)
        path = str(tmp_path / f"{config}.npz")
        loader.save_native(path, config, weights.synth_state(spec, seed=seed))
        return path
    return make


@pytest.mark.parametrize("config,kw", [("single_tech", dict(coverage=30)),
                                       ("hybrid_ensemble2", dict(coverage=30, hybrid_coverage=15))], ids=["single_tech", "hybrid_ensemble2_meta_on_reference"])
def test_shared_call_is_bit_identical_to_the_direct_per_site_call(model_file, config, kw):
    """the first client starts the server (a fresh child process: this process's GPU state is not inherited, the
    client itself never uses the GPU); with ONE client every launch holds one site, exactly like the direct call: identical keys,
    identical 5-tuples bit for bit, for a single-expert model and for two technologies + meta weights read off the reference segment."""
    path = model_file(config)
    direct = loader.load(path, providePredictions=True)
    shared_net = loader.load(path, shared=True, providePredictions=True)
    try:
        assert shared_net.info["n_experts"] == direct.engine.n_experts and shared_net.info["has_meta"] == direct.engine.has_meta
        for fd, seg in _calls(config, 24, 5, **kw):
            want, want_meta, want_keys = _as_arrays(direct(fd, seg))
            got, got_meta, got_keys = _as_arrays(shared_net(fd, seg))
            assert got_keys == want_keys and np.array_equal(got, want) and np.array_equal(got_meta, want_meta)
        shared_net.providePredictions = False
        fd, seg = _calls(config, 1, 6, **kw)[0]
        direct.providePredictions = False
        assert {k: float(v) for k, v in shared_net(fd, seg).items()} == {k: float(v) for k, v in direct(fd, seg).items()}
        stats = shared_net.server_stats()
        # two engines for a single-chain model (their launches run out of phase), one for a model whose small launches run lanes
        assert stats["sites"] == 25 and stats["launches"] == 25 and stats["errors"] == 0 and stats["engines"] == (2 if config == "single_tech" else 1)
        # a second "worker" finds the running server instead of starting one
        again = loader.load(path, shared=True, connect_only=True)
        assert again.info["pid"] == shared_net.info["pid"] and again.info["slot"] != shared_net.info["slot"]
        again.close()
    finally:
        shared_net.close()
        direct.close()


def _worker(path, rendezvous, rank, n_calls, out_q):
    os.environ["HELLO_SHARED_DIR"] = rendezvous
    os.environ["HELLO_SHARED_IDLE_EXIT"] = "1"
    sys.path.insert(0, ROOT)
    import torch
    torch.set_num_threads(1)
    from hello_amd import loader as ld
    assert not torch.cuda.is_initialized()
    net = ld.load(path, shared=True, providePredictions=True)
    calls = _calls("single_tech", 16, 40 + rank, coverage=30)
    for fd, seg in calls[:4]:
        net(fd, seg)
    got = [_as_arrays(net(*calls[i % len(calls)]))[0] for i in range(n_calls)]
    stats = net.server_stats()
    used_gpu = torch.cuda.is_initialized()
    net.close()
    out_q.put((rank, got, stats, used_gpu))


def test_worker_processes_share_one_server_and_their_sites_share_launches(model_file, tmp_path):
    """Six single-threaded worker processes, each with the reference's loop (load the model, one site per call): ONE server process
    scores them all, in launches of several sites; every answer equals the direct per-site call's to ~1e-6 (a site's answer depends
    on its launch's composition at the 1e-7 level); the workers never initialise the GPU."""
    path = model_file("single_tech")
    direct = loader.load(path, providePredictions=True)
    n_workers, n_calls = 6, 120
    want = {r: [_as_arrays(direct(fd, seg))[0] for fd, seg in _calls("single_tech", 16, 40 + r, coverage=30)] for r in range(n_workers)}
    direct.close()
    ctx = mp.get_context("spawn")                 # fresh interpreters: nothing of this process's GPU state is inherited
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(path, os.environ["HELLO_SHARED_DIR"], r, n_calls, q)) for r in range(n_workers)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    worst, pids = 0.0, set()
    for rank, got, stats, used_gpu in results:
        assert not used_gpu
        for i, g in enumerate(got):
            w = want[rank][i % 16]
            assert g.shape == w.shape
            worst = max(worst, float(np.abs(g - w).max()))
    assert worst <= 1e-6, worst
    stats = max((r[2] for r in results), key=lambda s: s["sites"])
    assert stats["clients_seen"] == n_workers and stats["errors"] == 0
    assert stats["largest_launch"] >= 2 and stats["launches"] < stats["sites"], stats
    print(f"shared server: {stats}; worst |d posterior| vs the direct per-site call {worst:.2e}")
