"""hello_amd.shared on the CPU: slot packing, coalescing and ordering across client processes, server death (the client raises,
no hang), client death (the slot comes back), refusals.  The server here is the product's native server (csrc/site_server.hip
behind its C ABI) with a deterministic Python scorer plugged into the ABI's scorer callback in place of an engine (a test double
for the GPU: the product's ``serve_model`` adds real engines and has no other path); the real thing is tests/test_gpu_shared.py."""
import multiprocessing as mp
import os
import signal
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hello_amd import shared, synth                                  # noqa: E402

INFO1 = dict(window=150, channels0=6, channels1=0, n_experts=1, has_meta=False, uses_ref=False, ensemble=False, arithmetic="fp32")
INFO3 = dict(window=150, channels0=6, channels1=6, n_experts=3, has_meta=True, uses_ref=True, ensemble=True, arithmetic="fp32")


def fake_scorer(n_experts, has_meta, delay=0.0, log=None):
    """A deterministic stand-in for Engine.forward: every output of a site is a function of that site's bytes alone, so the answer a
    client must get does not depend on which sites shared its launch."""
    def score(reads0, rpa0, aps, reads1, rpa1, ref):
        if delay:
            time.sleep(delay)
        if log is not None:
            log.append(int(aps.shape[0]))
        r_off = np.concatenate([[0], np.cumsum(rpa0)])
        per_allele = np.array([float(reads0[r_off[a]:r_off[a + 1]].sum(dtype=np.int64) % 9973) for a in range(rpa0.shape[0])], np.float32)
        if reads1 is not None:
            r1 = np.concatenate([[0], np.cumsum(rpa1)])
            per_allele += np.array([float(reads1[r1[a]:r1[a + 1]].sum(dtype=np.int64) % 7919) for a in range(rpa1.shape[0])], np.float32)
        logits = np.stack([per_allele / 1000 - 5 + e for e in range(n_experts)]).astype(np.float32)
        a_off = np.concatenate([[0], np.cumsum(aps)])
        meta = None
        if has_meta:
            shift = [0.0 if ref is None else (int(ref[s].sum(dtype=np.int64)) % 5) / 100 for s in range(aps.shape[0])]
            meta = np.array([[0.5 + d, 0.3 - d, 0.2] for d in shift], np.float32)
        post = []
        for s in range(aps.shape[0]):
            a = int(aps[s])
            base = per_allele[a_off[s]:a_off[s + 1]]
            pairs = [(i, j) for i in range(a) for j in range(i, a)]
            post.append(np.array([[r + (base[i] + 2 * base[j]) / 1e5 for i, j in pairs] for r in range(4)], np.float32))
        return logits, meta, np.concatenate(post, axis=1)
    return score


def expected(info, feature_dict, segment):
    """What a client must receive for one site: the fake scorer on that site alone."""
    from hello_amd.wrapper import ScoringNetwork
    reads0, rpa0, reads1, rpa1, aps, ref, names = ScoringNetwork._pack([(feature_dict, segment)], need_ref=bool(info["uses_ref"]))
    if not info["channels1"]:
        reads1 = rpa1 = None
    return fake_scorer(info["n_experts"], info["has_meta"])(reads0, rpa0, aps, reads1, rpa1, ref if info["uses_ref"] else None)


def _serve(sock, shm, info, scorers, delay, idle, max_clients, ready):
    server = shared.SiteServer(sock, shm, info, scorers=[fake_scorer(info["n_experts"], info["has_meta"], delay) for _ in range(scorers)],
                               slot_bytes=1 << 20, max_clients=max_clients, idle_exit_s=idle)
    signal.signal(signal.SIGTERM, lambda *_: server.stop())
    ready.set()
    server.serve()


@pytest.fixture
def server(tmp_path):
    """-> start(info, scorers=2, delay=0, idle=None, max_clients=16) -> (socket path, process); every server is torn down afterwards."""
    ctx = mp.get_context("fork")
    started = []

    def start(info, scorers=2, delay=0.0, idle=None, max_clients=16):
        sock, shm = str(tmp_path / f"s{len(started)}.sock"), str(tmp_path / f"s{len(started)}.slots")
        ready = ctx.Event()
        proc = ctx.Process(target=_serve, args=(sock, shm, info, scorers, delay, idle, max_clients, ready), daemon=True)
        proc.start()
        assert ready.wait(30)
        started.append(proc)
        return sock, proc
    yield start
    for proc in started:
        if proc.is_alive():
            proc.terminate()
        proc.join(10)
        if proc.is_alive():
            proc.kill()


def _sites(n, seed, hybrid=False):
    sys.path.insert(0, ROOT)
    import bench
    b = synth.make_sites(n, seed=seed, coverage=12, **({"hybrid_coverage": 6} if hybrid else {}))
    return bench.feature_dicts(b)


def test_slot_layout_round_trips_a_site():
    """What the client writes into a slot is what the server's views read back: both technologies, counts, the reference segment;
    the result area comes back shaped [E, A] / [3] / [4, A(A+1)/2]; a site that does not fit is refused with the limit named."""
    lay = shared.SlotLayout(150, 6, 7, 1 << 20)
    buf = bytearray(2 * lay.slot_bytes)
    slot = shared._Slot(buf, 1, lay)
    rng = np.random.default_rng(0)
    rpa0, rpa1 = np.array([3, 1, 5], np.int32), np.array([2, 2, 1], np.int32)
    reads0 = rng.integers(0, 255, (9, 150, 6), dtype=np.uint8)
    reads1 = rng.integers(0, 255, (5, 150, 7), dtype=np.uint8)
    ref = rng.integers(0, 2, (1, 150, 5), dtype=np.uint8)
    slot.write_site(reads0, rpa0, reads1, rpa1, ref)
    h = slot.header
    assert (int(h[shared.H_ALLELES]), int(h[shared.H_READS0]), int(h[shared.H_READS1]), int(h[shared.H_HAS_REF]), int(h[shared.H_PAIRS])) == (3, 9, 5, 1, 6)
    assert np.array_equal(slot.rpa0[:3], rpa0) and np.array_equal(slot.rpa1[:3], rpa1) and np.array_equal(slot.ref.reshape(150, 5), ref[0])
    n0, n1 = reads0.size, reads1.size
    assert np.array_equal(slot.reads[:n0].reshape(reads0.shape), reads0) and np.array_equal(slot.reads[n0:n0 + n1].reshape(reads1.shape), reads1)
    assert not bytes(buf[:lay.slot_bytes]).strip(b"\0")                                 # nothing spilled into slot 0
    # the result area as the server fills it: rows of fixed stride (logits [3][64], posteriors [4][2080])
    logits, meta, post = rng.standard_normal((3, 3)).astype(np.float32), np.array([.2, .3, .5], np.float32), rng.random((4, 6)).astype(np.float32)
    slot.logits.reshape(3, shared.MAX_ALLELES)[:, :3] = logits
    slot.meta[:3] = meta
    slot.post.reshape(4, shared.MAX_PAIRS)[:, :6] = post
    got = slot.read_result(3, True)
    assert np.array_equal(got[0], logits) and np.array_equal(got[1], meta) and np.array_equal(got[2], post)
    with pytest.raises(ValueError, match="does not fit a shared slot"):
        slot.write_site(np.zeros((1200, 150, 6), np.uint8), np.array([1200], np.int32), None, None, None)
    with pytest.raises(ValueError, match="at most 64"):
        slot.write_site(np.zeros((65, 150, 6), np.uint8), np.ones(65, np.int32), None, None, None)


def test_one_client_gets_the_reference_structures(server):
    """``network(featureDict, ref_segment)`` through the shared server returns what the per-site surface returns: a dict over the
    unordered allele pairs in first-seen product order holding 0-dim tensors, or the 5-tuple with providePredictions
    (MixtureOfExpertsAdvanced.py:562-589; single-expert models answer meta [1, 0, 0])."""
    import torch
    sock, _ = server(INFO1)
    with shared.SharedScoringNetwork("unused", socket_path=sock) as net:
        assert net.eval() is net and net.train(False) is net and net.info["slot"] == 0
        for fd, seg in _sites(5, 3):
            want_logits, _, want_post = expected(INFO1, fd, seg)
            out = net(fd, seg)
            names = list(fd)
            assert list(out) == [(names[i], names[j]) for i in range(len(names)) for j in range(i, len(names))]
            assert all(isinstance(v, torch.Tensor) and v.dim() == 0 for v in out.values())
            assert np.array_equal(np.array([float(v) for v in out.values()], np.float32), want_post[0])
        net.providePredictions = True
        fd, seg = _sites(1, 4)[0]
        mix, e0, e1, e2, meta = net(fd, seg)
        want = expected(INFO1, fd, seg)[2]
        for row, got in enumerate((mix, e0, e1, e2)):
            assert np.array_equal(np.array([float(v) for v in got.values()], np.float32), want[row])
        assert meta.tolist() == [1.0, 0.0, 0.0]
        stats = net.server_stats()
        assert stats["sites"] == 6 and stats["launches"] == 6 and stats["largest_launch"] == 1 and stats["clients"] == 1 and stats["engines"] == 2
        # inputs the engine would refuse are refused in the client, before anything is submitted
        with pytest.raises(ValueError, match="expected \\[rows, 150, 6\\]"):
            net({"A": (np.zeros((2, 150, 7), np.uint8), None)}, seg)
        with pytest.raises(ValueError, match="integers in 0..255"):
            net({"A": (np.full((2, 150, 6), 0.5, np.float32), None)}, seg)
        assert net.server_stats()["sites"] == 6
        # a client is another process: a header that lies is refused by the server's checks (reason in the slot), not followed
        good = net(fd, seg)
        for field, value in ((shared.H_READS0, 10 ** 6), (shared.H_ALLELES, 0), (shared.H_ALLELES, 65), (shared.H_READS1, 3)):
            keep = int(net._slot.header[field])
            net._slot.header[field] = value
            assert net._roundtrip() == shared.ERR and "describes no site that fits" in net._slot.read_error()
            net._slot.header[field] = keep
        net._slot.rpa0[0] += 1                                                            # reads per allele no longer add up
        assert net._roundtrip() == shared.ERR
        net._slot.rpa0[0] -= 1
        assert net._roundtrip() == shared.OK and net(fd, seg)[0].keys() == good[0].keys()  # the server is none the worse for it
    with pytest.raises(RuntimeError, match="closed"):
        net(fd, seg)


def _client(sock, info, seed, calls, hybrid, out_q):
    try:
        net = shared.SharedScoringNetwork("unused", socket_path=sock, providePredictions=True)
        sites = _sites(8, seed, hybrid)
        bad = 0
        for i in range(calls):
            fd, seg = sites[i % len(sites)]
            mix, e0, e1, e2, meta = net(fd, seg)
            want_logits, want_meta, want_post = expected(info, fd, seg)
            got = np.array([[float(v) for v in row.values()] for row in (mix, e0, e1, e2)], np.float32)
            bad += int(not np.array_equal(got, want_post))
            if info["has_meta"]:
                bad += int(not np.array_equal(meta.numpy(), want_meta[0]))
        net.close()
        out_q.put((seed, bad))
    except Exception as exc:                       # noqa: BLE001
        out_q.put((seed, repr(exc)))


@pytest.mark.parametrize("info,hybrid", [(INFO1, False), (INFO3, True)], ids=["single_tech", "hybrid_three_experts_meta_ref"])
def test_concurrent_clients_are_coalesced_and_each_gets_its_own_answer(server, info, hybrid):
    """Eight worker processes, one site per call each, against one server whose launches take 3 ms: every call returns ITS site's
    answer (ordering: a launch's results are scattered back to the slots they came from), and the server scored them in launches
    of several sites (K blocked workers -> launches of up to K sites, never more launches than sites)."""
    sock, _ = server(info, scorers=2, delay=0.003)
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_client, args=(sock, info, 100 + k, 40, hybrid, q)) for k in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(r[0] for r in results) == list(range(100, 108)) and all(r[1] == 0 for r in results), results
    with shared.SharedScoringNetwork("unused", socket_path=sock) as probe:
        stats = probe.server_stats()
    assert stats["sites"] == 8 * 40 and stats["clients_seen"] == 9 and stats["errors"] == 0
    assert stats["launches"] < stats["sites"] and 2 <= stats["largest_launch"] <= 8, stats


def test_server_death_raises_in_the_client_and_does_not_hang(server):
    """a server killed while a worker's site is in flight -- the worker's call raises RuntimeError promptly (its
    blocked recv sees the closed socket); later calls raise too; nothing waits for a dead process."""
    sock, proc = server(INFO1, scorers=1, delay=5.0)
    net = shared.SharedScoringNetwork("unused", socket_path=sock, request_timeout=60.0)
    fd, seg = _sites(1, 5)[0]
    killer = mp.get_context("fork").Process(target=lambda: (time.sleep(0.5), os.kill(proc.pid, signal.SIGKILL)))
    killer.start()
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match="went away"):
        net(fd, seg)
    assert time.monotonic() - t0 < 4.0                                  # not the scorer's 5 s, not the 60 s timeout
    killer.join()
    with pytest.raises(RuntimeError, match="closed"):
        net(fd, seg)
    with pytest.raises(RuntimeError, match="no scoring server accepts|hung up during the handshake"):
        shared.SharedScoringNetwork("unused", socket_path=sock)                           # (the dying process's backlog may still take a connect)


def test_a_wedged_server_is_bounded_by_the_request_timeout(server):
    sock, _ = server(INFO1, scorers=1, delay=3.0)
    net = shared.SharedScoringNetwork("unused", socket_path=sock, request_timeout=0.5)
    fd, seg = _sites(1, 6)[0]
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match="did not answer within"):
        net(fd, seg)
    assert time.monotonic() - t0 < 2.5


def _die_mid_call(sock):
    from hello_amd.wrapper import ScoringNetwork
    net = shared.SharedScoringNetwork("unused", socket_path=sock)
    reads0, rpa0, _, _, _, _, _ = ScoringNetwork._pack(_sites(1, 7)[:1], need_ref=False)
    net._slot.write_site(reads0, rpa0, None, None, None)
    net._sock.sendall(shared.REQ)
    os._exit(0)                                    # gone while its site is queued / in flight


def test_client_death_frees_its_slot(server):
    """Two slots only: clients that die (one of them with a request in flight) give their slots back, so later workers connect; a
    third simultaneous client is refused by name."""
    sock, _ = server(INFO1, scorers=1, delay=0.2, max_clients=2)
    ctx = mp.get_context("fork")
    for _ in range(3):
        p = ctx.Process(target=_die_mid_call, args=(sock,))
        p.start()
        p.join(30)
    time.sleep(0.6)
    a = shared.SharedScoringNetwork("unused", socket_path=sock)
    b = shared.SharedScoringNetwork("unused", socket_path=sock)
    assert {a.info["slot"], b.info["slot"]} == {0, 1}
    with pytest.raises(RuntimeError, match="all 2 slots are taken"):
        shared.SharedScoringNetwork("unused", socket_path=sock)
    fd, seg = _sites(1, 8)[0]
    assert np.array_equal(np.array([float(v) for v in a(fd, seg).values()], np.float32), expected(INFO1, fd, seg)[2][0])
    a.close()
    time.sleep(0.7)
    c = shared.SharedScoringNetwork("unused", socket_path=sock)
    assert c.info["slot"] == a.info["slot"]
    b.close()
    c.close()


def test_idle_server_leaves_and_cleans_up(server, tmp_path):
    sock, proc = server(INFO1, idle=0.6)
    with shared.SharedScoringNetwork("unused", socket_path=sock) as net:
        shm = net.info["shm_path"]
        assert os.path.exists(shm) and os.path.exists(sock)
        time.sleep(1.0)                                                  # a connected client keeps it alive
        assert proc.is_alive()
    proc.join(10)
    assert not proc.is_alive() and not os.path.exists(sock) and not os.path.exists(shm)


def test_rendezvous_is_per_model_file_and_device(tmp_path):
    a = tmp_path / "a.npz"
    a.write_bytes(b"x" * 10)
    p0 = shared.rendezvous_paths(str(a), 0, str(tmp_path / "rv"))
    assert p0 == shared.rendezvous_paths(str(a), 0, str(tmp_path / "rv")) and p0 != shared.rendezvous_paths(str(a), 1, str(tmp_path / "rv"))
    link = tmp_path / "link.npz"
    link.symlink_to(a)
    assert shared.rendezvous_paths(str(link), 0, str(tmp_path / "rv")) == p0               # the same file under another name
    time.sleep(0.01)
    a.write_bytes(b"y" * 11)                                                               # a re-trained model: its own server
    assert shared.rendezvous_paths(str(a), 0, str(tmp_path / "rv")) != p0
    assert oct(os.stat(tmp_path / "rv").st_mode & 0o777) == "0o700"


def test_auto_device_spreads_workers_over_the_gpus(monkeypatch):
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(os, "getpid", lambda: 4243)
    assert shared.pick_device("auto") == 4243 % 8 and shared.pick_device(3) == 3 and shared.pick_device("5") == 5
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)                             # no GPU visible: device 0 (the server will say so)
    assert shared.pick_device("auto") == 0


def test_load_shared_refuses_what_load_refuses_and_never_falls_back_to_the_cpu(tmp_path, monkeypatch):
    """``loader.load(path, shared=True)``: a git-LFS pointer is refused by name before any server starts; with a real model file and
    no GPU the server child exits (``Engine()`` raises: there is no CPU fallback) and the client raises with the end of its log."""
    from hello_amd import loader, netspec as ns, weights
    monkeypatch.setenv("HELLO_SHARED_DIR", str(tmp_path / "rv"))
    lfs = tmp_path / "model.wrapper.dnn"
    lfs.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:abc\nsize 12341951\n")
    with pytest.raises(ValueError, match="git-LFS pointer"):
        loader.load(str(lfs), shared=True)
    assert not os.path.exists(tmp_path / "rv") or not os.listdir(tmp_path / "rv")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the shared server starts for real (tests/test_gpu_shared.py)")
    spec = ns.build("single_tech")
    path = str(tmp_path / "single.npz")
    loader.save_native(path, "single_tech", weights.synth_state(spec, seed=1))
    with pytest.raises(RuntimeError, match="exited with status .* before it was ready"):
        loader.load(path, shared=True, start_timeout=240)


def test_native_server_is_clean_under_threadsanitizer(tmp_path):
    """The server's source compiled as plain host C++ with tests/abi/site_server_tsan.cpp under -fsanitize=thread: ten client THREADS
    speaking the wire protocol (hanging up and reconnecting in the middle of their runs) against three scorer threads with launch
    grouping on -- every answer is the caller's own, every site is scored exactly once in launches of several sites, and
    ThreadSanitizer reports no data race (leader / follower hand-over, in-flight / zombie slots, the fd table, statistics)."""
    import shutil
    import subprocess
    cxx = shutil.which("g++") or shutil.which("c++")
    assert cxx, "no C++ compiler on this box"
    exe = str(tmp_path / "site_server_tsan")
    build = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-Wall", "-I", os.path.join(ROOT, "include"), "-x", "c++",
                            os.path.join(ROOT, "hello_amd", "csrc", "site_server.hip"), os.path.join(ROOT, "tests", "abi", "site_server_tsan.cpp"),
                            "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:exitcode=66"))
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-4000:]
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr[-2000:])
    fields = dict(zip(run.stdout.split()[0::2], run.stdout.split()[1::2]))
    assert fields["sites"] == "3000" and fields["failures"] == "0" and fields["errors"] == "0" and int(fields["launches"]) < 3000
    assert int(fields["clients_seen"]) > 10                                               # the clients really hung up and came back
