"""Cross-process site coalescing behind the UNCHANGED per-site call.

The reference's deployment form is a pool of single-threaded worker processes, each of which loads the model and calls
``network(featureDict, ref_segment)`` once per site (python/call.py:111,215-221; python/caller_calling.py:863-868,872-891).
Given one engine each, K such workers put K contexts on the card that time-slice nine launches per site.  Here the workers
keep their loop and their call, and share ONE engine process per GPU instead:

    network = hello_amd.loader.load(path, shared=True)         # in every worker; the first one starts the server
    network.eval(); network.providePredictions = True
    out = network(featureDict, ref_segment)                    # packs the site into a shared-memory slot and blocks

* ``SharedScoringNetwork`` (the client; never touches the GPU): validates and packs a site exactly like
  ``ScoringNetwork._pack``, writes its bytes and counts into its slot of a ``/dev/shm`` segment, sends one byte on a
  Unix-domain socket and blocks in ``recv`` for the one-byte answer; then reads logits / meta / pair posteriors out of the
  slot and shapes them like the reference's return value.
* ``SiteServer`` (one per (model file, GPU); a fresh child process of the first client -- ``python -m hello_amd.shared
  --serve ...`` started with ``subprocess.Popen`` before anything in it touches the GPU; never a re-exec): the NATIVE server of
  the in-tree library (``csrc/site_server.hip`` behind ``hello_site_server_*`` of ``include/hello_mi355x.h``): one scorer
  thread per engine, leader / follower -- the idle thread waits on the sockets itself, drains ALL pending slots into one
  ``hello_engine_forward`` launch, lingers a few tens of microseconds for the workers that could still send a site, and
  scatters the answers back; no interpreter on a site's path.  K blocked workers become launches of up to K sites instead
  of K one-site launches.  A site's answer does not depend on which other sites shared its launch beyond the engine's
  documented ~1e-6 (bit-identical when it was alone).  This module is the client, the rendezvous and
  the server process's ``main``.

Liveness: the socket is the liveness signal both ways.  A server that dies closes every client's socket: the blocked
``recv`` returns and the client raises ``RuntimeError`` (no hang; a timeout bounds even a wedged server).  A client that dies
frees its slot.  The server leaves ``idle_exit_s`` seconds after its last client has gone (the reference's workers load the
model once per shard, caller_calling.py:863: the server outlives them, so the model is built once, not once per shard).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import mmap
import os
import socket
import struct
import sys
import threading
import time
from typing import Callable, Dict, Optional, Sequence

import numpy as np

PROTOCOL = 1
MAX_ALLELES = 64                                   # alleles of one site a slot can hold
MAX_PAIRS = MAX_ALLELES * (MAX_ALLELES + 1) // 2
DEFAULT_SLOT_BYTES = 4 << 20                       # 1 000 reads x 250 x 7 = 1.75 MB fit with room to spare
DEFAULT_MAX_CLIENTS = 64
HEADER_INTS = 16
H_ALLELES, H_READS0, H_READS1, H_HAS_REF, H_PAIRS, H_ERRLEN = range(6)
ERR_BYTES = 1024
REQ, STATS, OK, ERR = b"R", b"S", b"K", b"E"


class SlotLayout:
    """Byte offsets inside one slot.  Fixed-size header and tables first, the result area, then the pileup bytes of both read
    technologies (whatever is left of the slot)."""

    def __init__(self, window: int, channels0: int, channels1: int, slot_bytes: int = DEFAULT_SLOT_BYTES):
        self.window, self.channels0, self.channels1, self.slot_bytes = int(window), int(channels0), int(channels1), int(slot_bytes)
        at = 0

        def take(n):
            nonlocal at
            here = at
            at = (at + n + 63) & ~63
            return here
        self.header = take(4 * HEADER_INTS)
        self.rpa0 = take(4 * MAX_ALLELES)
        self.rpa1 = take(4 * MAX_ALLELES)
        self.ref = take(self.window * 5)
        self.logits = take(4 * 3 * MAX_ALLELES)
        self.meta = take(4 * 4)
        self.post = take(4 * 4 * MAX_PAIRS)
        self.err = take(ERR_BYTES)
        self.reads = at
        self.read_capacity = self.slot_bytes - at
        if self.read_capacity < self.window * max(self.channels0, 1):
            raise ValueError(f"slots of {slot_bytes} bytes cannot hold one read of this model")

    def row_bytes(self, tech: int) -> int:
        return self.window * (self.channels0 if tech == 0 else self.channels1)


class _Slot:
    """NumPy views of one slot of the mapped segment: the client's side (the server reads and writes the same bytes in C++)."""

    def __init__(self, buf, index: int, lay: SlotLayout):
        base = index * lay.slot_bytes
        self.index, self.lay = index, lay
        u8 = np.frombuffer(buf, dtype=np.uint8, count=lay.slot_bytes, offset=base)
        self.u8 = u8
        self.header = u8[lay.header:lay.header + 4 * HEADER_INTS].view(np.int32)
        self.rpa0 = u8[lay.rpa0:lay.rpa0 + 4 * MAX_ALLELES].view(np.int32)
        self.rpa1 = u8[lay.rpa1:lay.rpa1 + 4 * MAX_ALLELES].view(np.int32)
        self.ref = u8[lay.ref:lay.ref + lay.window * 5]
        self.logits = u8[lay.logits:lay.logits + 4 * 3 * MAX_ALLELES].view(np.float32)
        self.meta = u8[lay.meta:lay.meta + 16].view(np.float32)
        self.post = u8[lay.post:lay.post + 16 * MAX_PAIRS].view(np.float32)
        self.err = u8[lay.err:lay.err + ERR_BYTES]
        self.reads = u8[lay.reads:]

    # -- client side -----------------------------------------------------------------------------------------------
    def write_site(self, reads0, rpa0, reads1, rpa1, ref) -> None:
        lay = self.lay
        a = int(rpa0.shape[0])
        n0 = int(reads0.shape[0]) * lay.row_bytes(0)
        n1 = 0 if reads1 is None else int(reads1.shape[0]) * lay.row_bytes(1)
        if a > MAX_ALLELES:
            raise ValueError(f"a site of {a} alleles does not fit a shared slot (at most {MAX_ALLELES}); score it through a private engine")
        if n0 + n1 > lay.read_capacity:
            raise ValueError(f"a site of {n0 + n1} pileup bytes does not fit a shared slot ({lay.read_capacity} bytes; "
                             f"HELLO_SHARED_SLOT_BYTES sizes the slots when the server starts)")
        self.reads[:n0] = reads0.reshape(-1)
        if n1:
            self.reads[n0:n0 + n1] = reads1.reshape(-1)
        self.rpa0[:a] = rpa0
        if rpa1 is not None:
            self.rpa1[:a] = rpa1
        if ref is not None:
            self.ref[:] = ref.reshape(-1)
        h = self.header
        h[H_ALLELES], h[H_READS0], h[H_READS1], h[H_HAS_REF] = a, reads0.shape[0], 0 if reads1 is None else reads1.shape[0], int(ref is not None)
        h[H_PAIRS] = a * (a + 1) // 2

    def read_result(self, n_experts: int, has_meta: bool):
        a, p = int(self.header[H_ALLELES]), int(self.header[H_PAIRS])
        logits = self.logits.reshape(3, MAX_ALLELES)[:n_experts, :a].copy()          # rows of fixed stride: the server scatters a whole launch at once
        meta = self.meta[:3].copy() if has_meta else None
        post = self.post.reshape(4, MAX_PAIRS)[:, :p].copy()
        return logits, meta, post

    def read_error(self) -> str:
        n = int(self.header[H_ERRLEN])
        return bytes(self.err[:max(0, min(n, ERR_BYTES))]).decode("utf-8", "replace")


def _send_msg(sock, obj) -> None:
    raw = json.dumps(obj).encode()
    sock.sendall(struct.pack("<I", len(raw)) + raw)


def _recv_exact(sock, n: int) -> bytes:
    out = b""
    while len(out) < n:
        chunk = sock.recv(n - len(out))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        out += chunk
    return out


def _recv_msg(sock):
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    if n > (1 << 20):
        raise ConnectionError(f"handshake message of {n} bytes")
    return json.loads(_recv_exact(sock, n).decode())


# ------------------------------------------------------------------------------------------------------------------
# server
# ------------------------------------------------------------------------------------------------------------------
Scorer = Callable[[np.ndarray, np.ndarray, np.ndarray, Optional[np.ndarray], Optional[np.ndarray], Optional[np.ndarray]], tuple]


class _ServerConfig(C.Structure):                  # hello_site_server_config of include/hello_mi355x.h
    _fields_ = [("window", C.c_int32), ("channels0", C.c_int32), ("channels1", C.c_int32), ("n_experts", C.c_int32),
                ("has_meta", C.c_int32), ("uses_ref", C.c_int32), ("max_clients", C.c_int32), ("max_batch_sites", C.c_int32),
                ("group_launches", C.c_int32), ("reserved", C.c_int32), ("slot_bytes", C.c_int64), ("idle_exit_s", C.c_double), ("linger_s", C.c_double), ("info_json", C.c_char_p)]


class _ServerStats(C.Structure):                   # hello_site_server_stats
    _fields_ = [("launches", C.c_int64), ("sites", C.c_int64), ("errors", C.c_int64), ("largest_launch", C.c_int32), ("clients_seen", C.c_int32)]


class _SlotLayoutC(C.Structure):                   # hello_site_slot_layout
    _fields_ = [(k, C.c_int64) for k in ("header", "rpa0", "rpa1", "ref", "logits", "meta", "post", "err", "reads", "read_capacity")]


_SCORER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                         C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32)
_server_lib = None


def server_library():
    """The in-tree library's server entry points (host C++, csrc/site_server.hip); loads without a GPU."""
    global _server_lib
    if _server_lib is None:
        from . import engine
        lib = engine.load_library()
        if not hasattr(lib, "hello_site_server_create"):
            raise RuntimeError(f"{engine._LIB_PATH} was built before the shared scoring server (hello_site_server_*): rebuild it")
        vp = C.c_void_p
        lib.hello_site_slot_layout_of.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(_SlotLayoutC)]
        lib.hello_site_server_create.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(_ServerConfig), C.POINTER(vp)]
        lib.hello_site_server_add_engine.argtypes = [vp, vp]
        lib.hello_site_server_add_scorer.argtypes = [vp, _SCORER_FN, vp]
        lib.hello_site_server_run.argtypes = [vp]
        lib.hello_site_server_stop.argtypes = [vp]
        lib.hello_site_server_stop.restype = None
        lib.hello_site_server_get_stats.argtypes = [vp, C.POINTER(_ServerStats)]
        lib.hello_site_server_destroy.argtypes = [vp]
        lib.hello_site_server_destroy.restype = None
        for fn in ("hello_site_slot_layout_of", "hello_site_server_create", "hello_site_server_add_engine", "hello_site_server_add_scorer",
                   "hello_site_server_run", "hello_site_server_get_stats"):
            getattr(lib, fn).restype = C.c_int
        _server_lib = lib
    return _server_lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"hello_mi355x: {server_library().hello_last_error().decode()} (status {rc})")


def _callback(score: Scorer, info: Dict):
    """A Python scorer behind the C ABI's hello_site_scorer (tests: the server without a GPU).  -> the ctypes callback object."""
    window, c0, c1, n_experts = info["window"], info["channels0"], info["channels1"], info["n_experts"]

    def view(ptr, dtype, shape):
        n = int(np.prod(shape))
        if not ptr or not n:
            return None if not ptr else np.zeros(shape, dtype)
        return np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype).reshape(shape)

    def fn(ctx, reads0, rpa0, reads1, rpa1, aps, ref, S, A, R0, R1, logits, meta, post, err, err_cap):
        try:
            counts = view(aps, np.int32, (S,))
            P = int((counts.astype(np.int64) * (counts + 1) // 2).sum())
            lg, mt, po = score(view(reads0, np.uint8, (R0, window, c0)), view(rpa0, np.int32, (A,)), counts,
                               view(reads1, np.uint8, (R1, window, c1)) if reads1 else None, view(rpa1, np.int32, (A,)) if rpa1 else None,
                               view(ref, np.uint8, (S, window, 5)) if ref else None)
            view(logits, np.float32, (n_experts, A))[...] = lg
            if meta and mt is not None:
                view(meta, np.float32, (S, 3))[...] = mt
            view(post, np.float32, (4, P))[...] = po
            return 0
        except Exception as exc:                   # noqa: BLE001 -- reported to the launch's clients through the C ABI
            raw = f"{type(exc).__name__}: {exc}".encode("utf-8", "replace")[:max(err_cap - 1, 0)]
            C.memmove(err, raw + b"\0", len(raw) + 1)
            return 1
    return _SCORER_FN(fn)


class SiteServer:
    """One scoring server: the native server of csrc/site_server.hip (a listening Unix socket, a shared-memory segment of
    ``max_clients`` slots, one leader / follower scorer thread per engine) behind its C ABI.  ``engines`` are ``Engine`` objects
    (the product: one ``hello_engine_forward`` per launch, no interpreter on the path); ``scorers`` are Python callables
    ``score(reads0, rpa0, aps, reads1, rpa1, ref) -> (logits [E, A], meta [S, 3] | None, posteriors [4, P])`` behind the ABI's
    scorer callback (tests drive the same server without a GPU through them).  ``info`` describes the model to clients: window,
    channels0, channels1, n_experts, has_meta, uses_ref (+ whatever else the handshake should carry: ensemble, arithmetic).

    A launch costs nearly the same for 1 or 16 sites (0.27 / 0.40 ms), so the thread that polls lingers up to ``linger_s`` for the
    clients that could still send a site (connected minus in flight elsewhere) before its launch goes out, and with several
    engines a launch takes its share of the clients (clients / engines) so that the groups run out of phase (``HELLO_SHARED_GROUPS=0``
    switches that off).  Measured on one MI355X, 16 workers: no lingering 24.8 k sites/s, lingering 27-30 k, + groups 31.2 k
    (profiles/r06_per_site_shared_native.txt; the round's Python prototype of this server: 17.7 k / 26.2 k, and groups were a loss
    there -- 22.3 k: its launches' host halves serialised on the interpreter lock; profiles/r06_per_site_shared_sweep.txt)."""

    def __init__(self, socket_path: str, shm_path: str, info: Dict, scorers: Sequence[Scorer] = (), engines: Sequence = (),
                 slot_bytes: int = DEFAULT_SLOT_BYTES, max_clients: int = DEFAULT_MAX_CLIENTS, idle_exit_s: Optional[float] = 15.0,
                 max_batch_sites: int = 4096, linger_s: Optional[float] = None):
        self.lib = server_library()
        self.socket_path, self.shm_path = socket_path, shm_path
        linger = float(linger_s if linger_s is not None else float(os.environ.get("HELLO_SHARED_LINGER_US", 120)) * 1e-6)
        extra = {k: v for k, v in info.items() if k not in ("window", "channels0", "channels1", "n_experts", "has_meta", "uses_ref")}
        self._info_json = json.dumps(extra)[1:-1].encode()
        cfg = _ServerConfig(int(info["window"]), int(info["channels0"]), int(info["channels1"]), int(info["n_experts"]), int(bool(info["has_meta"])),
                            int(bool(info["uses_ref"])), int(max_clients), int(max_batch_sites),
                            int(os.environ.get("HELLO_SHARED_GROUPS", "1") != "0"), 0, int(slot_bytes),
                            -1.0 if idle_exit_s is None else float(idle_exit_s), linger, self._info_json or None)
        handle = C.c_void_p()
        _check(self.lib.hello_site_server_create(socket_path.encode(), shm_path.encode(), C.byref(cfg), C.byref(handle)))
        self.handle = handle
        self._keep = [_callback(s, info) for s in scorers]          # the callback objects must outlive the server
        self._engines = list(engines)
        for cb in self._keep:
            _check(self.lib.hello_site_server_add_scorer(self.handle, cb, None))
        for e in self._engines:                     # the server's launches are a few sites each: the laned program of a multi-chain model
            _check(self.lib.hello_site_server_add_engine(self.handle, e.small_launch_handle()))
        self.n_scorers = len(self._keep) + len(self._engines)

    @property
    def stats(self) -> Dict:
        st = _ServerStats()
        _check(self.lib.hello_site_server_get_stats(self.handle, C.byref(st)))
        return dict(launches=int(st.launches), sites=int(st.sites), largest_launch=int(st.largest_launch), clients_seen=int(st.clients_seen),
                    errors=int(st.errors))

    def serve(self) -> None:
        """Run until ``stop()`` or until no client has been connected for ``idle_exit_s`` seconds; then release everything (the
        segment and the socket file are unlinked).  The blocking C call runs on a helper thread so that this (the main) thread
        keeps executing bytecode: Python signal handlers -- which call ``stop()`` -- run while the server serves."""
        result = []
        worker = threading.Thread(target=lambda: result.append(self.lib.hello_site_server_run(self.handle)), daemon=True)
        worker.start()
        try:
            while worker.is_alive():
                worker.join(0.2)
        finally:
            self.lib.hello_site_server_stop(self.handle)
            worker.join(15.0)
            self.last_stats = self.stats
            self.lib.hello_site_server_destroy(self.handle)
            self.handle = None
        if result and result[0] != 0:
            _check(result[0])

    def stop(self) -> None:
        if self.handle:
            self.lib.hello_site_server_stop(self.handle)


def model_info(program, spec) -> Dict:
    return dict(window=int(program.window), channels0=int(program.channels0), channels1=int(program.channels1),
                n_experts=int(program.n_experts), has_meta=bool(program.has_meta), uses_ref=bool(program.uses_ref),
                ensemble=bool(spec.ensemble), arithmetic=str(program.arithmetic))


def serve_model(path: str, device: int, socket_path: str, shm_path: str, engines: int = 0, arithmetic: Optional[str] = None,
                slot_bytes: int = DEFAULT_SLOT_BYTES, max_clients: int = DEFAULT_MAX_CLIENTS, idle_exit_s: float = 15.0) -> None:
    """The server process's main: load the model, build ``engines`` engines on ``device`` (no CPU fallback: this raises without the
    HIP library or a gfx950 device), then bind the socket -- a connectable socket means a ready server -- and serve.  ``engines`` 0 =
    two for a single-chain model (their launches run out of phase), one for a model whose small launches already run several lanes
    (16 workers, C4: 28.4 k sites/s with one engine, 26.1 k with two; hybrid_full 22.7 k / 19.9 k: the lanes' streams compete)."""
    import signal
    from . import loader
    from .engine import Engine
    spec, state = loader.load_spec(path)
    engs = [Engine(spec, state, device=device, arithmetic=arithmetic)]
    if engines <= 0:
        engines = 1 if engs[0].small_launch_handle() is engs[0].lanes_handle else 2
    engs += [Engine(spec, state, device=device, arithmetic=arithmetic) for _ in range(engines - 1)]
    server = SiteServer(socket_path, shm_path, model_info(engs[0].program, spec), engines=engs,
                        slot_bytes=slot_bytes, max_clients=max_clients, idle_exit_s=idle_exit_s)
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, lambda *_: server.stop())
    print(f"hello_amd.shared: serving {path} on cuda:{device} with {len(engs)} engine(s) at {socket_path} (pid {os.getpid()})", file=sys.stderr, flush=True)
    try:
        server.serve()
    finally:
        for e in engs:
            e.close()
        print(f"hello_amd.shared: server {os.getpid()} leaves: {json.dumps(getattr(server, 'last_stats', {}))}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------------------------
# client
# ------------------------------------------------------------------------------------------------------------------
def rendezvous_paths(path: str, device: int, directory: Optional[str] = None):
    """-> (socket path, shared-memory path, lock path, log path) of THE server of (model file, GPU, user): derived from the file's
    real path, size and modification time, so a re-trained model under the same name gets its own server."""
    real = os.path.realpath(path)
    st = os.stat(real)
    key = hashlib.sha1(f"{real}|{st.st_size}|{st.st_mtime_ns}|{device}|{PROTOCOL}".encode()).hexdigest()[:20]
    directory = directory or os.environ.get("HELLO_SHARED_DIR") or os.path.join("/tmp", f"hello_amd-{os.getuid()}")
    os.makedirs(directory, mode=0o700, exist_ok=True)
    shm_dir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else directory
    return (os.path.join(directory, key + ".sock"), os.path.join(shm_dir, f"hello_amd-{os.getuid()}-{key}.slots"),
            os.path.join(directory, key + ".lock"), os.path.join(directory, key + ".log"))


def _try_connect(socket_path: str, timeout: float):
    sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    sock.settimeout(timeout)
    try:
        sock.connect(socket_path)
    except OSError:
        sock.close()
        return None
    return sock


def start_server(path: str, device: int, paths, engines: int = 0, arithmetic: Optional[str] = None, idle_exit_s: float = 15.0,
                 start_timeout: float = 300.0):
    """Start the server of ``path`` as a fresh CHILD process (a new session: it outlives this client) and wait until its socket
    accepts connections.  The caller holds the rendezvous lock.  -> a connected socket."""
    import subprocess
    socket_path, shm_path, _, log_path = paths
    if os.path.exists(socket_path):
        os.unlink(socket_path)                     # a stale file: nobody accepted on it (checked by the caller, under the lock)
    cmd = [sys.executable, "-m", "hello_amd.shared", "--serve", "--model", os.path.realpath(path), "--device", str(device),
           "--socket", socket_path, "--shm", shm_path, "--engines", str(engines), "--idle-exit", str(idle_exit_s),
           "--slot-bytes", str(int(os.environ.get("HELLO_SHARED_SLOT_BYTES", DEFAULT_SLOT_BYTES)))]
    if arithmetic:
        cmd += ["--arithmetic", arithmetic]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    with open(log_path, "ab") as log:
        child = subprocess.Popen(cmd, stdin=subprocess.DEVNULL, stdout=log, stderr=log, start_new_session=True, env=env, cwd=root)
    deadline = time.monotonic() + start_timeout
    while time.monotonic() < deadline:
        sock = _try_connect(socket_path, 5.0) if os.path.exists(socket_path) else None
        if sock is not None:
            return sock
        rc = child.poll()
        if rc is not None:
            tail = ""
            try:
                tail = open(log_path, "r", errors="replace").read()[-1500:]
            except OSError:
                pass
            raise RuntimeError(f"the scoring server for {path} exited with status {rc} before it was ready (there is no CPU fallback: it "
                               f"needs the HIP library and a gfx950 device); its log {log_path} ends:\n{tail}")
        time.sleep(0.05)
    child.terminate()
    raise RuntimeError(f"the scoring server for {path} was not ready after {start_timeout:.0f} s (log: {log_path})")


def pick_device(device) -> int:
    """``device`` as given, or -- ``"auto"`` -- this worker's share of the node: process id modulo the number of GPUs (a pool's workers
    have consecutive ids, so they spread evenly; counting devices does not initialise the GPU), one scoring server per GPU."""
    if device != "auto":
        return int(device)
    import torch
    return os.getpid() % max(torch.cuda.device_count(), 1)


class SharedScoringNetwork:
    """The object a worker holds as ``network`` when it loaded the model with ``shared=True``: the per-site plug-in surface of
    ``hello_amd.wrapper.ScoringNetwork`` (``.eval()``, ``.providePredictions``, ``__call__(featureDict, ref_segment)`` with the
    reference's return structures, MixtureOfExpertsAdvanced.py:520-589), scored by the shared server of (model file, GPU)."""

    def __init__(self, path: str, device=0, providePredictions: bool = False, engines: Optional[int] = None, arithmetic: Optional[str] = None,
                 request_timeout: float = 120.0, start_timeout: float = 300.0, idle_exit_s: Optional[float] = None, directory: Optional[str] = None,
                 connect_only: bool = False, socket_path: Optional[str] = None):
        self.path, self.device = path, pick_device(device)
        self.providePredictions = providePredictions
        self.training = False
        self.request_timeout = float(request_timeout)
        self._sock = None
        if socket_path is not None:                # an explicit server (tests, an operator-run server)
            sock = _try_connect(socket_path, 5.0)
            if sock is None:
                raise RuntimeError(f"no scoring server accepts connections at {socket_path}")
        else:
            engines = int(engines or os.environ.get("HELLO_SHARED_ENGINES", 0))      # scorer threads of a server THIS client starts (0: by the model)
            idle_exit_s = float(idle_exit_s if idle_exit_s is not None else os.environ.get("HELLO_SHARED_IDLE_EXIT", 15.0))
            sock = self._connect_or_start(engines, arithmetic, start_timeout, idle_exit_s, directory, connect_only)
        try:
            _send_msg(sock, {"protocol": PROTOCOL, "pid": os.getpid()})
            info = _recv_msg(sock)
        except (OSError, ConnectionError, ValueError) as exc:
            sock.close()
            raise RuntimeError(f"the scoring server hung up during the handshake: {exc!r}") from exc
        if "error" in info:
            sock.close()
            raise RuntimeError(f"the scoring server refused this client: {info['error']}")
        self.info = info
        self.layout = SlotLayout(info["window"], info["channels0"], info["channels1"], info["slot_bytes"])
        fd = os.open(info["shm_path"], os.O_RDWR)
        try:
            self._map = mmap.mmap(fd, info["max_clients"] * self.layout.slot_bytes)
        finally:
            os.close(fd)
        self._slot = _Slot(self._map, int(info["slot"]), self.layout)
        sock.settimeout(self.request_timeout)
        self._sock = sock

    def _connect_or_start(self, engines, arithmetic, start_timeout, idle_exit_s, directory, connect_only):
        import fcntl
        paths = rendezvous_paths(self.path, self.device, directory)
        sock = _try_connect(paths[0], 5.0)
        if sock is not None:
            return sock
        if connect_only:
            raise RuntimeError(f"no scoring server for {self.path} on cuda:{self.device} is running")
        with open(paths[2], "a+") as lock:         # one starter: the others wait here and then find the socket
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                sock = _try_connect(paths[0], 5.0)
                if sock is not None:
                    return sock
                return start_server(self.path, self.device, paths, engines, arithmetic, idle_exit_s, start_timeout)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)

    # torch.nn.Module look-alikes the caller touches
    def eval(self):
        return self

    def train(self, mode: bool = False):
        if mode:
            raise NotImplementedError("inference-only engine")
        return self

    def close(self):
        if self._sock is not None:
            try:
                self._sock.close()
            finally:
                self._sock = None
        self._slot = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -- the call ---------------------------------------------------------------------------------------------------
    def _roundtrip(self):
        if self._sock is None:
            raise RuntimeError("this network is closed")
        try:
            self._sock.sendall(REQ)
            answer = self._sock.recv(1)
        except socket.timeout as exc:
            self.close()
            raise RuntimeError(f"the scoring server did not answer within {self.request_timeout:.0f} s") from exc
        except OSError as exc:
            self.close()
            raise RuntimeError(f"the scoring server went away: {exc!r}") from exc
        if not answer:
            self.close()
            raise RuntimeError("the scoring server went away (connection closed) while a site was being scored")
        return answer

    def __call__(self, featureDict, segment):
        import torch
        from .wrapper import ScoringNetwork, _SINGLE_EXPERT_META, pair_keys
        if self._sock is None:
            raise RuntimeError("this network is closed (or its scoring server went away): load the model again")
        info = self.info
        reads0, rpa0, reads1, rpa1, aps, ref, names = ScoringNetwork._pack([(featureDict, segment)], need_ref=bool(info["uses_ref"]))
        window = info["window"]
        if reads0.ndim != 3 or tuple(reads0.shape[1:]) != (window, info["channels0"]):
            raise ValueError(f"reads0: expected [rows, {window}, {info['channels0']}] (channels last), got {tuple(reads0.shape)}")
        if info["channels1"]:
            if reads1 is None:
                raise ValueError("this model scores two read technologies: every allele needs both tensors")
            if reads1.ndim != 3 or tuple(reads1.shape[1:]) != (window, info["channels1"]):
                raise ValueError(f"reads1: expected [rows, {window}, {info['channels1']}] (channels last), got {tuple(reads1.shape)}")
        else:
            reads1 = rpa1 = None                   # like the reference's single-technology forward, which never reads them
        if (rpa0 < 1).any() or (rpa1 is not None and (rpa1 < 1).any()):
            raise ValueError("every allele needs at least one read (the featurizer gives an unsupported allele one all-zero read)")
        if info["uses_ref"]:
            if ref is None:
                raise ValueError("this model reads the one-hot reference segment: ref_segment [1, window, 5] is required")
            if tuple(ref.shape[1:]) != (window, 5):
                raise ValueError(f"ref_segment: expected [1, {window}, 5], got {tuple(ref.shape)}")
        else:
            ref = None
        self._slot.write_site(reads0, rpa0, reads1, rpa1, ref)
        answer = self._roundtrip()
        if answer != OK:
            raise RuntimeError(f"hello_mi355x (shared server): {self._slot.read_error()}")
        logits, meta, post = self._slot.read_result(info["n_experts"], info["has_meta"])
        keys = pair_keys(names[0])
        n = len(keys)
        scalars = torch.from_numpy(post.reshape(-1)).unbind(0)           # 0-dim tensors of all four rows with one call
        rows = [dict(zip(keys, scalars[r * n:(r + 1) * n])) for r in range(4)]
        if not self.providePredictions:
            return rows[0]
        m = torch.from_numpy(meta) if info["has_meta"] else _SINGLE_EXPERT_META.clone()
        return rows[0], rows[1], rows[2], rows[3], m

    forward = __call__

    def server_stats(self) -> Dict:
        """The server's counters: launches, sites, largest_launch, clients_seen, errors, clients, engines."""
        if self._sock is None:
            raise RuntimeError("this network is closed")
        try:
            self._sock.sendall(STATS)
            answer = self._sock.recv(1)
        except OSError as exc:
            self.close()
            raise RuntimeError(f"the scoring server went away: {exc!r}") from exc
        if answer != OK:
            self.close()
            raise RuntimeError("the scoring server went away")
        return json.loads(self._slot.read_error())

    def score_sites(self, sites):
        """One result per site, in order (each site is its own request: the server batches across processes, not inside one)."""
        return [self(fd, seg) for fd, seg in sites]


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="the shared scoring server of one (model file, GPU); normally started by the first client")
    ap.add_argument("--serve", action="store_true", required=True)
    ap.add_argument("--model", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--socket", required=True)
    ap.add_argument("--shm", required=True)
    ap.add_argument("--engines", type=int, default=0)
    ap.add_argument("--arithmetic", default=None)
    ap.add_argument("--slot-bytes", type=int, default=DEFAULT_SLOT_BYTES)
    ap.add_argument("--max-clients", type=int, default=DEFAULT_MAX_CLIENTS)
    ap.add_argument("--idle-exit", type=float, default=15.0)
    a = ap.parse_args(argv)
    serve_model(a.model, a.device, a.socket, a.shm, a.engines, a.arithmetic, a.slot_bytes, a.max_clients, a.idle_exit)


if __name__ == "__main__":
    main()
