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
  --serve ...`` started with ``subprocess.Popen`` before anything in it touches the GPU; never a re-exec): one selector
  loop takes the request bytes, scorer threads -- one engine each -- drain ALL pending slots into one
  ``Engine.forward`` launch and scatter the answers back.  While one scorer's launch is on the GPU the next batch
  collects and is launched by the other, so K blocked workers become launches of up to K sites instead of K one-site
  launches.  Results are in slot order = arrival order; a site's answer does not depend on which other sites shared its
  launch beyond the engine's documented ~1e-6 (bit-identical when it was alone).

Liveness: the socket is the liveness signal both ways.  A server that dies closes every client's socket: the blocked
``recv`` returns and the client raises ``RuntimeError`` (no hang; a timeout bounds even a wedged server).  A client that dies
frees its slot.  The server leaves ``idle_exit_s`` seconds after its last client has gone (the reference's workers load the
model once per shard, caller_calling.py:863: the server outlives them, so the model is built once, not once per shard).
"""
from __future__ import annotations

import hashlib
import json
import mmap
import os
import socket
import struct
import sys
import threading
import time
from collections import deque
from typing import Callable, Dict, List, Optional, Sequence

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
    """NumPy views of one slot of the mapped segment."""

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
        logits = self.logits[:n_experts * a].reshape(n_experts, a).copy()
        meta = self.meta[:3].copy() if has_meta else None
        post = self.post[:4 * p].reshape(4, p).copy()
        return logits, meta, post

    def read_error(self) -> str:
        n = int(self.header[H_ERRLEN])
        return bytes(self.err[:max(0, min(n, ERR_BYTES))]).decode("utf-8", "replace")

    # -- server side -----------------------------------------------------------------------------------------------
    def site_views(self):
        """-> (alleles, reads0 [R0, L, C0] view, rpa0, reads1 view | None, rpa1 | None, ref [L, 5] | None); raises ValueError on a
        header that does not describe a site that fits the slot (a client is another process: nothing it writes is trusted)."""
        lay, h = self.lay, self.header
        a, r0, r1 = int(h[H_ALLELES]), int(h[H_READS0]), int(h[H_READS1])
        if not (1 <= a <= MAX_ALLELES) or r0 < a or r1 < 0 or (r1 and not lay.channels1):
            raise ValueError(f"slot {self.index}: header describes no site (alleles {a}, reads {r0} / {r1})")
        n0, n1 = r0 * lay.row_bytes(0), r1 * lay.row_bytes(1)
        if n0 + n1 > lay.read_capacity:
            raise ValueError(f"slot {self.index}: {n0 + n1} pileup bytes exceed the slot")
        rpa0 = self.rpa0[:a]
        rpa1 = self.rpa1[:a] if r1 else None
        if int(rpa0.sum()) != r0 or rpa0.min() < 1 or (rpa1 is not None and (int(rpa1.sum()) != r1 or rpa1.min() < 1)):
            raise ValueError(f"slot {self.index}: reads per allele do not add up to the read count (or an allele has no read)")
        reads0 = self.reads[:n0].reshape(r0, lay.window, lay.channels0)
        reads1 = self.reads[n0:n0 + n1].reshape(r1, lay.window, lay.channels1) if r1 else None
        ref = self.ref.reshape(lay.window, 5) if h[H_HAS_REF] else None
        return a, reads0, rpa0, reads1, rpa1, ref

    def write_result(self, logits, meta, post) -> None:
        self.logits[:logits.size] = logits.reshape(-1)
        if meta is not None:
            self.meta[:3] = meta
        self.post[:post.size] = post.reshape(-1)

    def write_error(self, message: str) -> None:
        raw = message.encode("utf-8", "replace")[:ERR_BYTES]
        self.err[:len(raw)] = np.frombuffer(raw, np.uint8)
        self.header[H_ERRLEN] = len(raw)


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


class SiteServer:
    """One scoring server: a listening Unix socket, a shared-memory segment of ``max_clients`` slots, one selector loop and
    ``len(scorers)`` scorer threads.  ``scorers`` are callables ``score(reads0, rpa0, aps, reads1, rpa1, ref) -> (logits [E, A],
    meta [S, 3] | None, posteriors [4, P])`` -- ``engine_scorer(Engine)`` in the product; each is called from its own thread only.
    ``info`` describes the model to clients: window, channels0, channels1, n_experts, has_meta, uses_ref, ensemble."""

    def __init__(self, socket_path: str, shm_path: str, info: Dict, scorers: Sequence[Scorer], slot_bytes: int = DEFAULT_SLOT_BYTES,
                 max_clients: int = DEFAULT_MAX_CLIENTS, idle_exit_s: Optional[float] = 15.0, max_batch_sites: int = 4096):
        self.socket_path, self.shm_path = socket_path, shm_path
        self.info = dict(info, protocol=PROTOCOL, slot_bytes=int(slot_bytes), max_clients=int(max_clients), shm_path=shm_path, pid=os.getpid())
        self.layout = SlotLayout(info["window"], info["channels0"], info["channels1"], slot_bytes)
        self.scorers = list(scorers)
        self.idle_exit_s, self.max_batch_sites = idle_exit_s, int(max_batch_sites)
        self.max_clients = int(max_clients)
        fd = os.open(shm_path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
        try:
            os.ftruncate(fd, self.max_clients * self.layout.slot_bytes)
            self._map = mmap.mmap(fd, self.max_clients * self.layout.slot_bytes)
        finally:
            os.close(fd)
        self.slots = [_Slot(self._map, i, self.layout) for i in range(self.max_clients)]
        self._free = deque(range(self.max_clients))
        self._socks: Dict[int, socket.socket] = {}             # slot index -> client socket
        self._pending: deque = deque()
        self._inflight: set = set()                            # slots a scorer thread is reading / writing right now
        self._zombies: set = set()                             # ... whose client went away meanwhile: freed when the launch is over
        self._cond = threading.Condition()
        self._stop = False
        self.stats = dict(launches=0, sites=0, largest_launch=0, clients_seen=0, errors=0)
        if os.path.exists(socket_path):
            os.unlink(socket_path)
        self._listener = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self._listener.bind(socket_path)
        os.chmod(socket_path, 0o600)
        self._listener.listen(self.max_clients)

    # -- scorer threads ---------------------------------------------------------------------------------------------
    def _reply(self, index: int, byte: bytes) -> None:
        sock = self._socks.get(index)
        if sock is None:
            return
        try:
            sock.sendall(byte)
        except OSError:
            pass                                   # the client went away while its site was being scored

    def _score_batch(self, score: Scorer, take: List[int]) -> None:
        sites, good = [], []
        for index in take:
            try:
                sites.append(self.slots[index].site_views())
                good.append(index)
            except ValueError as exc:
                self.slots[index].write_error(str(exc))
                self._reply(index, ERR)
        if not good:
            return
        second = sites[0][3] is not None
        with_ref = sites[0][5] is not None
        # a launch holds sites of one shape of call (every client of a server speaks for the same model): a stray one is refused
        keep = [k for k, s in enumerate(sites) if (s[3] is not None) == second and (s[5] is not None) == with_ref]
        for k in set(range(len(sites))) - set(keep):
            self.slots[good[k]].write_error("this site's optional inputs (second technology / reference segment) differ from the launch's")
            self._reply(good[k], ERR)
        sites, good = [sites[k] for k in keep], [good[k] for k in keep]
        aps = np.array([s[0] for s in sites], np.int32)
        cat = (lambda k: sites[0][k]) if len(sites) == 1 else (lambda k: np.concatenate([s[k] for s in sites]))
        reads0, rpa0 = cat(1), cat(2)
        reads1, rpa1 = (cat(3), cat(4)) if second else (None, None)
        ref = np.stack([s[5] for s in sites]) if with_ref else None
        try:
            logits, meta, post = score(reads0, rpa0, aps, reads1, rpa1, ref)
        except Exception as exc:                   # the whole launch failed: every site of it is answered with the reason
            self.stats["errors"] += 1
            for index in good:
                self.slots[index].write_error(f"{type(exc).__name__}: {exc}")
                self._reply(index, ERR)
            return
        a_off = np.concatenate([[0], np.cumsum(aps)])
        p_off = np.concatenate([[0], np.cumsum(aps.astype(np.int64) * (aps + 1) // 2)])
        for k, index in enumerate(good):
            self.slots[index].write_result(logits[:, a_off[k]:a_off[k + 1]], None if meta is None else meta[k], post[:, p_off[k]:p_off[k + 1]])
            self._reply(index, OK)
        self.stats["launches"] += 1
        self.stats["sites"] += len(good)
        self.stats["largest_launch"] = max(self.stats["largest_launch"], len(good))

    def _scorer_loop(self, score: Scorer) -> None:
        while True:
            with self._cond:
                while not self._pending and not self._stop:
                    self._cond.wait(0.25)
                if self._stop and not self._pending:
                    return
                take = [self._pending.popleft() for _ in range(min(len(self._pending), self.max_batch_sites))]
                self._inflight.update(take)
            try:
                self._score_batch(score, take)
            finally:
                with self._cond:
                    self._inflight.difference_update(take)
                    for index in [i for i in take if i in self._zombies]:
                        self._zombies.discard(index)
                        self._free.append(index)

    # -- selector loop ----------------------------------------------------------------------------------------------
    def _accept(self, sel) -> None:
        import selectors
        conn, _ = self._listener.accept()
        try:
            conn.settimeout(5.0)
            hello = _recv_msg(conn)
            if hello.get("protocol") != PROTOCOL:
                _send_msg(conn, {"error": f"protocol {hello.get('protocol')} != {PROTOCOL}"})
                conn.close()
                return
            if not self._free:
                _send_msg(conn, {"error": f"all {self.max_clients} slots are taken"})
                conn.close()
                return
            index = self._free.popleft()
            _send_msg(conn, dict(self.info, slot=index))
            conn.settimeout(None)
        except (OSError, ValueError, ConnectionError):
            conn.close()
            return
        self._socks[index] = conn
        self.stats["clients_seen"] += 1
        sel.register(conn, selectors.EVENT_READ, index)

    def _drop(self, sel, index: int) -> None:
        sock = self._socks.pop(index, None)
        if sock is not None:
            try:
                sel.unregister(sock)
            except (KeyError, ValueError):
                pass
            sock.close()
            with self._cond:                       # a dead client's queued site is not scored; its slot is reusable afterwards
                try:
                    self._pending.remove(index)
                except ValueError:
                    pass
                if index in self._inflight:        # ... but not while a launch still reads it
                    self._zombies.add(index)
                else:
                    self._free.append(index)

    def serve(self) -> None:
        """Run until ``stop()`` or until no client has been connected for ``idle_exit_s`` seconds.  Cleans up its files."""
        import selectors
        sel = selectors.DefaultSelector()
        sel.register(self._listener, selectors.EVENT_READ, None)
        threads = [threading.Thread(target=self._scorer_loop, args=(s,), daemon=True) for s in self.scorers]
        for t in threads:
            t.start()
        idle_since = time.monotonic()
        try:
            while not self._stop:
                events = sel.select(timeout=0.5)
                for key, _ in events:
                    if key.data is None:
                        self._accept(sel)
                        continue
                    index = key.data
                    try:
                        data = key.fileobj.recv(64)
                    except OSError:
                        data = b""
                    if not data:
                        self._drop(sel, index)
                        continue
                    if data.count(REQ):            # one outstanding request per client: further bytes are ignored
                        with self._cond:
                            self._pending.append(index)
                            self._cond.notify()
                    elif data.count(STATS):        # the server's counters, as JSON in the slot's message area
                        self.slots[index].write_error(json.dumps(dict(self.stats, clients=len(self._socks), engines=len(self.scorers))))
                        self._reply(index, OK)
                if self._socks:
                    idle_since = time.monotonic()
                elif self.idle_exit_s is not None and time.monotonic() - idle_since > self.idle_exit_s:
                    break
        finally:
            self._stop = True
            with self._cond:
                self._cond.notify_all()
            for t in threads:
                t.join(timeout=5.0)
            for index in list(self._socks):
                self._drop(sel, index)
            sel.close()
            self._listener.close()
            for path in (self.socket_path, self.shm_path):
                try:
                    os.unlink(path)
                except OSError:
                    pass

    def stop(self) -> None:
        self._stop = True


def engine_scorer(engine) -> Scorer:
    """The product's scorer: one ``Engine.forward`` launch over host arrays (logits, meta, pair posteriors back on the host)."""
    def score(reads0, rpa0, aps, reads1, rpa1, ref):
        return engine.forward(reads0, rpa0, aps, reads1, rpa1, ref, posteriors=True)
    return score


def model_info(program, spec) -> Dict:
    return dict(window=int(program.window), channels0=int(program.channels0), channels1=int(program.channels1),
                n_experts=int(program.n_experts), has_meta=bool(program.has_meta), uses_ref=bool(program.uses_ref),
                ensemble=bool(spec.ensemble), arithmetic=str(program.arithmetic))


def serve_model(path: str, device: int, socket_path: str, shm_path: str, engines: int = 2, arithmetic: Optional[str] = None,
                slot_bytes: int = DEFAULT_SLOT_BYTES, max_clients: int = DEFAULT_MAX_CLIENTS, idle_exit_s: float = 15.0) -> None:
    """The server process's main: load the model, build ``engines`` engines on ``device`` (no CPU fallback: this raises without the
    HIP library or a gfx950 device), then bind the socket -- a connectable socket means a ready server -- and serve."""
    import signal
    from . import loader
    from .engine import Engine
    spec, state = loader.load_spec(path)
    engs = [Engine(spec, state, device=device, arithmetic=arithmetic) for _ in range(max(1, engines))]
    server = SiteServer(socket_path, shm_path, model_info(engs[0].program, spec), [engine_scorer(e) for e in engs],
                        slot_bytes=slot_bytes, max_clients=max_clients, idle_exit_s=idle_exit_s)
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, lambda *_: server.stop())
    print(f"hello_amd.shared: serving {path} on cuda:{device} with {len(engs)} engine(s) at {socket_path} (pid {os.getpid()})", file=sys.stderr, flush=True)
    try:
        server.serve()
    finally:
        for e in engs:
            e.close()
        print(f"hello_amd.shared: server {os.getpid()} leaves: {json.dumps(server.stats)}", file=sys.stderr, flush=True)


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


def start_server(path: str, device: int, paths, engines: int = 2, arithmetic: Optional[str] = None, idle_exit_s: float = 15.0,
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


class SharedScoringNetwork:
    """The object a worker holds as ``network`` when it loaded the model with ``shared=True``: the per-site plug-in surface of
    ``hello_amd.wrapper.ScoringNetwork`` (``.eval()``, ``.providePredictions``, ``__call__(featureDict, ref_segment)`` with the
    reference's return structures, MixtureOfExpertsAdvanced.py:520-589), scored by the shared server of (model file, GPU)."""

    def __init__(self, path: str, device: int = 0, providePredictions: bool = False, engines: Optional[int] = None, arithmetic: Optional[str] = None,
                 request_timeout: float = 120.0, start_timeout: float = 300.0, idle_exit_s: float = 15.0, directory: Optional[str] = None,
                 connect_only: bool = False, socket_path: Optional[str] = None):
        self.path, self.device = path, int(device)
        self.providePredictions = providePredictions
        self.training = False
        self.request_timeout = float(request_timeout)
        self._sock = None
        if socket_path is not None:                # an explicit server (tests, an operator-run server)
            sock = _try_connect(socket_path, 5.0)
            if sock is None:
                raise RuntimeError(f"no scoring server accepts connections at {socket_path}")
        else:
            engines = int(engines or os.environ.get("HELLO_SHARED_ENGINES", 2))      # scorer threads of a server THIS client starts
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
        rows = [dict(zip(keys, torch.from_numpy(post[r]).unbind(0))) for r in range(4)]
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
    ap.add_argument("--engines", type=int, default=2)
    ap.add_argument("--arithmetic", default=None)
    ap.add_argument("--slot-bytes", type=int, default=DEFAULT_SLOT_BYTES)
    ap.add_argument("--max-clients", type=int, default=DEFAULT_MAX_CLIENTS)
    ap.add_argument("--idle-exit", type=float, default=15.0)
    a = ap.parse_args(argv)
    serve_model(a.model, a.device, a.socket, a.shm, a.engines, a.arithmetic, a.slot_bytes, a.max_clients, a.idle_exit)


if __name__ == "__main__":
    main()
