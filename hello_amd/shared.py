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
        logits = self.logits.reshape(3, MAX_ALLELES)[:n_experts, :a].copy()          # rows of fixed stride: the server scatters a whole launch at once
        meta = self.meta[:3].copy() if has_meta else None
        post = self.post.reshape(4, MAX_PAIRS)[:, :p].copy()
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
        self.logits.reshape(3, MAX_ALLELES)[:logits.shape[0], :logits.shape[1]] = logits
        if meta is not None:
            self.meta[:3] = meta
        self.post.reshape(4, MAX_PAIRS)[:, :post.shape[1]] = post

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
    """One scoring server: a listening Unix socket, a shared-memory segment of ``max_clients`` slots and ``len(scorers)`` scorer
    threads.  ``scorers`` are callables ``score(reads0, rpa0, aps, reads1, rpa1, ref) -> (logits [E, A], meta [S, 3] | None,
    posteriors [4, P])`` -- ``engine_scorer(Engine)`` in the product; each is called from its own thread only.  ``info`` describes
    the model to clients: window, channels0, channels1, n_experts, has_meta, uses_ref, ensemble.

    Threads are leader / follower: whichever scorer thread is idle holds the poll lock, waits on the sockets itself (accepting
    clients, noticing the dead ones, reading request bytes), takes EVERY pending slot as its launch and hands the lock to the next
    idle thread before it scores -- no hand-over between a poller and a scorer on a site's way in.  A poller that has fewer requests
    than there are clients who could still send one (connected minus in flight elsewhere) lingers up to ``linger_s`` for them: a
    launch costs nearly the same for 1 or 16 sites (0.27 / 0.40 ms), so a few tens of microseconds of patience buy sites per launch
    (measured on one MI355X, 16 workers: 17.7 k sites/s without lingering, 26.2 k with 120 us; cutting the clients into one group per
    engine so that the groups run out of phase was measured too and is slower -- 22.3 k: the launches' host halves serialise on the
    interpreter lock; profiles/r06_per_site_shared_sweep.txt)."""

    def __init__(self, socket_path: str, shm_path: str, info: Dict, scorers: Sequence[Scorer], slot_bytes: int = DEFAULT_SLOT_BYTES,
                 max_clients: int = DEFAULT_MAX_CLIENTS, idle_exit_s: Optional[float] = 15.0, max_batch_sites: int = 4096,
                 linger_s: Optional[float] = None):
        self.socket_path, self.shm_path = socket_path, shm_path
        self.info = dict(info, protocol=PROTOCOL, slot_bytes=int(slot_bytes), max_clients=int(max_clients), shm_path=shm_path, pid=os.getpid())
        self.layout = lay = SlotLayout(info["window"], info["channels0"], info["channels1"], slot_bytes)
        self.scorers = list(scorers)
        self.idle_exit_s, self.max_batch_sites = idle_exit_s, int(max_batch_sites)
        self.linger_s = float(linger_s if linger_s is not None else float(os.environ.get("HELLO_SHARED_LINGER_US", 120)) * 1e-6)
        self.max_clients = m = int(max_clients)
        fd = os.open(shm_path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
        try:
            os.ftruncate(fd, m * lay.slot_bytes)
            self._map = mmap.mmap(fd, m * lay.slot_bytes)
        finally:
            os.close(fd)
        self.slots = [_Slot(self._map, i, lay) for i in range(m)]
        # the same fields of EVERY slot as one strided array each: a launch gathers and scatters with a few NumPy calls, whatever its size
        sb = lay.slot_bytes
        view = lambda shape, dtype, off, strides: np.ndarray(shape, dtype, buffer=self._map, offset=off, strides=(sb,) + strides)   # noqa: E731
        self._hdr = view((m, HEADER_INTS), np.int32, lay.header, (4,))
        self._rpa0 = view((m, MAX_ALLELES), np.int32, lay.rpa0, (4,))
        self._rpa1 = view((m, MAX_ALLELES), np.int32, lay.rpa1, (4,))
        self._ref = view((m, lay.window, 5), np.uint8, lay.ref, (5, 1))
        self._logits = view((m, 3, MAX_ALLELES), np.float32, lay.logits, (4 * MAX_ALLELES, 4))
        self._meta = view((m, 4), np.float32, lay.meta, (4,))
        self._post = view((m, 4, MAX_PAIRS), np.float32, lay.post, (4 * MAX_PAIRS, 4))
        self._free = deque(range(m))
        self._socks: Dict[int, socket.socket] = {}             # slot index -> client socket
        self._pending: List[int] = []                          # requests read off the sockets, not yet part of a launch (poller's)
        self._inflight: set = set()                            # slots a scorer thread is reading / writing right now
        self._zombies: set = set()                             # ... whose client went away meanwhile: freed when the launch is over
        self._state = threading.Lock()                         # guards _inflight / _zombies / _free / stats
        self._poll = threading.Lock()                          # the leader's: sockets, selector, _pending, _socks
        self._stop = False
        self._idle_since = time.monotonic()
        self._sel = None
        self.stats = dict(launches=0, sites=0, largest_launch=0, clients_seen=0, errors=0)
        if os.path.exists(socket_path):
            os.unlink(socket_path)
        self._listener = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self._listener.bind(socket_path)
        os.chmod(socket_path, 0o600)
        self._listener.listen(m)

    # -- a launch ---------------------------------------------------------------------------------------------------
    def _reply(self, index: int, byte: bytes) -> None:
        sock = self._socks.get(index)
        if sock is None:
            return
        try:
            sock.sendall(byte)
        except OSError:
            pass                                   # the client went away while its site was being scored

    def _refuse(self, index: int, message: str) -> None:
        self.slots[index].write_error(message)
        self._reply(index, ERR)

    def _gather(self, take: List[int]):
        """The slots of one launch -> (slot indices kept, batch arrays).  A client is another process: every header is checked
        against the slot's capacity and its own tables before a byte of it is followed; a slot that fails is answered with the reason
        and left out."""
        lay = self.layout
        idx = np.asarray(take, dtype=np.int64)
        h = self._hdr[idx]
        a, r0, r1, has_ref = (h[:, k].astype(np.int64) for k in (H_ALLELES, H_READS0, H_READS1, H_HAS_REF))
        rb0, rb1 = lay.row_bytes(0), lay.row_bytes(1)
        cols = np.arange(MAX_ALLELES)[None, :]
        live = cols < a[:, None]
        t0 = np.where(live, self._rpa0[idx], 0)
        t1 = np.where(live, self._rpa1[idx], 0)
        ok = (a >= 1) & (a <= MAX_ALLELES) & (r0 >= a) & (r1 >= 0) & ((r1 == 0) | (lay.channels1 > 0)) & (r0 * rb0 + r1 * rb1 <= lay.read_capacity)
        ok &= (t0.sum(axis=1) == r0) & (np.where(live, t0, 1).min(axis=1) >= 1)
        ok &= (r1 == 0) | ((t1.sum(axis=1) == r1) & (np.where(live, t1, 1).min(axis=1) >= 1))
        # a launch holds sites of one shape of call (every client of a server speaks for the same model): a stray one is refused
        second, with_ref = bool(r1[0] > 0), bool(has_ref[0])
        ok &= ((r1 > 0) == second) & ((has_ref != 0) == with_ref)
        if not ok.all():
            for k in np.nonzero(~ok)[0]:
                self._refuse(int(idx[k]), f"slot {int(idx[k])}: the header describes no site that fits the slot and the launch (alleles {int(a[k])}, "
                                          f"reads {int(r0[k])} / {int(r1[k])}; reads per allele must add up, every allele needs a read, optional "
                                          f"inputs must match the launch's)")
            keep = np.nonzero(ok)[0]
            if keep.size == 0:
                return [], None
            idx, a, r0, r1, t0, t1, live = idx[keep], a[keep], r0[keep], r1[keep], t0[keep], t1[keep], live[keep]
        n = idx.shape[0]
        slots = self.slots
        if n == 1:
            i = int(idx[0])
            reads0 = slots[i].reads[:int(r0[0]) * rb0]
            reads1 = slots[i].reads[int(r0[0]) * rb0:int(r0[0]) * rb0 + int(r1[0]) * rb1] if second else None
        else:
            reads0 = np.concatenate([slots[int(i)].reads[:int(n0)] for i, n0 in zip(idx, r0 * rb0)])
            reads1 = np.concatenate([slots[int(i)].reads[int(n0):int(n0 + n1)] for i, n0, n1 in zip(idx, r0 * rb0, r1 * rb1)]) if second else None
        batch = dict(aps=a.astype(np.int32), rpa0=t0[live].astype(np.int32), reads0=reads0.reshape(-1, lay.window, lay.channels0),
                     rpa1=t1[live].astype(np.int32) if second else None,
                     reads1=reads1.reshape(-1, lay.window, lay.channels1) if second else None,
                     ref=np.ascontiguousarray(self._ref[idx]) if with_ref else None)
        return idx, batch

    def _scatter(self, idx, aps, logits, meta, post) -> None:
        n = idx.shape[0]
        a = aps.astype(np.int64)
        site_a = np.repeat(np.arange(n), a)
        local_a = np.arange(site_a.shape[0]) - np.repeat(np.cumsum(a) - a, a)
        self._logits[idx[site_a][None, :], np.arange(logits.shape[0])[:, None], local_a[None, :]] = logits
        if meta is not None:
            self._meta[idx, :3] = meta
        p = a * (a + 1) // 2
        site_p = np.repeat(np.arange(n), p)
        local_p = np.arange(site_p.shape[0]) - np.repeat(np.cumsum(p) - p, p)
        self._post[idx[site_p][None, :], np.arange(4)[:, None], local_p[None, :]] = post

    def _score_batch(self, score: Scorer, take: List[int]) -> None:
        idx, b = self._gather(take)
        if b is None:
            return
        try:
            logits, meta, post = score(b["reads0"], b["rpa0"], b["aps"], b["reads1"], b["rpa1"], b["ref"])
            self._scatter(idx, b["aps"], logits, meta, post)
        except Exception as exc:                   # the whole launch failed: every site of it is answered with the reason
            with self._state:
                self.stats["errors"] += 1
            for index in idx:
                self._refuse(int(index), f"{type(exc).__name__}: {exc}")
            return
        for index in idx:
            self._reply(int(index), OK)
        with self._state:
            self.stats["launches"] += 1
            self.stats["sites"] += int(idx.shape[0])
            self.stats["largest_launch"] = max(self.stats["largest_launch"], int(idx.shape[0]))

    # -- the leader: sockets -----------------------------------------------------------------------------------------
    def _accept(self) -> None:
        import selectors
        conn, _ = self._listener.accept()
        try:
            conn.settimeout(5.0)
            hello = _recv_msg(conn)
            if hello.get("protocol") != PROTOCOL:
                _send_msg(conn, {"error": f"protocol {hello.get('protocol')} != {PROTOCOL}"})
                conn.close()
                return
            with self._state:
                index = self._free.popleft() if self._free else None
            if index is None:
                _send_msg(conn, {"error": f"all {self.max_clients} slots are taken"})
                conn.close()
                return
            _send_msg(conn, dict(self.info, slot=index))
            conn.settimeout(None)
        except (OSError, ValueError, ConnectionError):
            conn.close()
            return
        self._socks[index] = conn
        with self._state:
            self.stats["clients_seen"] += 1
        self._sel.register(conn, selectors.EVENT_READ, index)

    def _drop(self, index: int) -> None:
        sock = self._socks.pop(index, None)
        if sock is None:
            return
        try:
            self._sel.unregister(sock)
        except (KeyError, ValueError):
            pass
        sock.close()
        if index in self._pending:                 # a dead client's queued site is not scored
            self._pending.remove(index)
        with self._state:                          # its slot is reusable -- but not while a launch still reads it
            if index in self._inflight:
                self._zombies.add(index)
            else:
                self._free.append(index)

    def _poll_once(self, timeout: float) -> None:
        for key, _ in self._sel.select(timeout=timeout):
            if key.data is None:
                self._accept()
                continue
            index = key.data
            try:
                data = key.fileobj.recv(64)
            except OSError:
                data = b""
            if not data:
                self._drop(index)
            elif data.count(REQ):                  # one outstanding request per client: further bytes are ignored
                self._pending.append(index)
            elif data.count(STATS):                # the server's counters, as JSON in the slot's message area
                with self._state:
                    stats = dict(self.stats, clients=len(self._socks), engines=len(self.scorers))
                self.slots[index].write_error(json.dumps(stats))
                self._reply(index, OK)

    def _collect(self) -> List[int]:
        """Called with the poll lock held: wait for requests, linger briefly for the clients that could still send one, and return
        the slots of the next launch (empty when stopping)."""
        first = None
        while not self._stop:
            if not self._pending:
                first = None
                self._poll_once(0.25)
                if self._socks or self._pending:
                    self._idle_since = time.monotonic()
                elif self.idle_exit_s is not None and time.monotonic() - self._idle_since > self.idle_exit_s:
                    self._stop = True
                continue
            now = time.monotonic()
            first = first if first is not None else now
            with self._state:
                could_still_come = len(self._socks) - len(self._inflight) - len(self._pending)
            left = self.linger_s - (now - first)
            if could_still_come <= 0 or left <= 0 or len(self._pending) >= self.max_batch_sites:
                break
            self._poll_once(min(left, 50e-6))
        n = self.max_batch_sites
        take, self._pending = self._pending[:n], self._pending[n:]
        with self._state:
            self._inflight.update(take)
        return take

    def _scorer_loop(self, score: Scorer) -> None:
        while not self._stop:
            with self._poll:
                take = self._collect() if not self._stop else []
            if not take:
                continue
            try:
                self._score_batch(score, take)
            finally:
                with self._state:
                    self._inflight.difference_update(take)
                    for index in [i for i in take if i in self._zombies]:
                        self._zombies.discard(index)
                        self._free.append(index)

    def serve(self) -> None:
        """Run until ``stop()`` or until no client has been connected for ``idle_exit_s`` seconds.  Cleans up its files."""
        import selectors
        self._sel = selectors.SelectSelector()     # select(2): microsecond timeouts (epoll's and poll's are rounded up to milliseconds)
        self._sel.register(self._listener, selectors.EVENT_READ, None)
        self._idle_since = time.monotonic()
        threads = [threading.Thread(target=self._scorer_loop, args=(s,), daemon=True) for s in self.scorers]
        try:
            for t in threads:
                t.start()
            while not self._stop and any(t.is_alive() for t in threads):
                time.sleep(0.05)
        finally:
            self._stop = True
            for t in threads:
                t.join(timeout=10.0)
            with self._poll:
                for index in list(self._socks):
                    self._drop(index)
                self._sel.close()
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
                 request_timeout: float = 120.0, start_timeout: float = 300.0, idle_exit_s: Optional[float] = None, directory: Optional[str] = None,
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
