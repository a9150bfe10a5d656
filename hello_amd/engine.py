"""ctypes binding of libhello_mi355x.so (include/hello_mi355x.h) and the batched operator surface.

``Engine.forward`` mirrors the reference's batched call
``dnn(tensors, numAllelesPerSite, numReadsPerAllele, reference_segments, *extra)``
(reference python/MixtureOfExpertsDNNFast.py:128-134 -> MixtureOfExpertsAdvanced.py:161).  There is no
CPU fallback: if the HIP library or a gfx950 device is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import compiler
from . import netspec as ns

_log = logging.getLogger(__name__)

# HELLO_LIB overrides the library path (kernel experiments); the default is the in-tree build
_LIB_PATH = os.environ.get("HELLO_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                        "libhello_mi355x.so")
HELLO_IN_DEVICE, HELLO_OUT_DEVICE, HELLO_LAYOUT_RCL = 1, 2, 4
ABI_VERSION = 2          # HELLO_ABI_VERSION of include/hello_mi355x.h
LANES_MAX_SITES = 64     # launches of at most this many sites run the laned program of a multi-chain model (Engine.forward).  ONE engine
                         # alone on a card gains from lanes at every size (hybrid_full: 1 site -34 %, 256 sites -22 %, 2 048 -7 %, 8 192
                         # -1.5 %; C4: -21 / -11 / -3 / -1 %), but several engines sharing a card (HostPipeline(engines=[...]), the
                         # 256-site small-batch form) LOSE beyond the latency regime -- their lanes' streams oversubscribe the hardware
                         # queues (four engines at 256 sites: C4 277 k -> 204 k sites/s).  Hence the default; Engine(lanes_max_sites=...)
                         # raises it for an engine that has the card to itself.

class HelloOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("domain", C.c_int32), ("src0", C.c_int32), ("src1", C.c_int32),
                ("dst", C.c_int32), ("res", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("lin", C.c_int32),
                ("lout", C.c_int32), ("flags", C.c_int32), ("seg", C.c_int32), ("c1", C.c_int32),
                ("a0", C.c_float), ("a1", C.c_float), ("w_off", C.c_int64), ("b_off", C.c_int64)]


class HelloBuffer(C.Structure):
    _fields_ = [("domain", C.c_int32), ("floats_per_row", C.c_int32)]


class HelloModelDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("window", C.c_int32), ("channels0", C.c_int32),
                ("channels1", C.c_int32), ("n_experts", C.c_int32), ("has_meta", C.c_int32),
                ("uses_ref", C.c_int32), ("n_buffers", C.c_int32), ("buffers", C.POINTER(HelloBuffer)),
                ("n_ops", C.c_int32), ("ops", C.POINTER(HelloOp))]


_lib = None


def load_library():
    """dlopen the in-tree HIP library (built by __graft_entry__.build() / hello_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` (there is no CPU fallback for the scoring path)")
    try:
        import torch  # noqa: F401  -- loads the HIP runtime torch ships, which the library then shares
    except Exception:
        pass
    lib = C.CDLL(_LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
    lib.hello_last_error.restype = C.c_char_p
    lib.hello_abi_version.restype = C.c_int
    # a stale library (HELLO_LIB override, a cached build) would read this layer's programs with another meaning of the
    # hello_op fields: refuse it here, before any program reaches it
    if lib.hello_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{_LIB_PATH} implements ABI version {lib.hello_abi_version()}, this Python layer speaks {ABI_VERSION}: "
                           f"rebuild it (`make -C hello_amd/csrc`)")
    lib.hello_engine_create.argtypes = [C.POINTER(HelloModelDesc), vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.hello_engine_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i64, f32p, f32p, f32p, i32, vp]
    lib.hello_engine_posteriors.argtypes = [vp, f32p, f32p, vp, i32, i32, i64, f32p, i32, vp]
    lib.hello_engine_synchronize.argtypes = [vp]
    lib.hello_engine_last_forward_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.hello_engine_set_profiling.argtypes = [vp, C.c_int]
    lib.hello_engine_op_times_ms.argtypes = [vp, C.POINTER(C.c_float), i32, C.POINTER(i32), C.POINTER(i32)]
    lib.hello_engine_set_profiling_filter.argtypes = [vp, i32]
    lib.hello_engine_debug_capture.argtypes = [vp, i32]
    lib.hello_engine_debug_read.argtypes = [vp, vp, i64, C.POINTER(i64)]
    # the kernel-timeline diagnostic is an addition within ABI version 2: a library built before it (HELLO_LIB pointing at an
    # older build, e.g. the baseline of an A/B run) still loads; record_stamps then says what is missing
    diagnostics = ("hello_engine_debug_stamps", "hello_engine_debug_read_stamps") if hasattr(lib, "hello_engine_debug_stamps") else ()
    if diagnostics:
        lib.hello_engine_debug_stamps.argtypes = [vp, i32]
        lib.hello_engine_debug_read_stamps.argtypes = [vp, vp, i64, C.POINTER(i64), C.POINTER(i32)]
    lib.hello_engine_stream.argtypes = [vp]
    lib.hello_engine_stream.restype = C.c_void_p
    lib.hello_engine_destroy.argtypes = [vp]
    lib.hello_engine_destroy.restype = None
    for fn in ("hello_engine_create", "hello_engine_forward", "hello_engine_posteriors",
               "hello_engine_synchronize", "hello_engine_last_forward_ms", "hello_engine_set_profiling",
               "hello_engine_op_times_ms", "hello_engine_set_profiling_filter", "hello_engine_debug_capture",
               "hello_engine_debug_read") + diagnostics:
        getattr(lib, fn).restype = C.c_int
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"hello_mi355x: {load_library().hello_last_error().decode()} (status {rc})")


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _i32_host(x) -> np.ndarray:
    if _is_torch(x):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=np.int32)


def n_pairs(alleles_per_site) -> int:
    a = np.asarray(alleles_per_site, dtype=np.int64)
    return int((a * (a + 1) // 2).sum())


def model_desc(program: "compiler.Program"):
    """-> (hello_model_desc of a compiled program, the ctypes arrays it points into -- keep them alive while it is used)."""
    ops = (HelloOp * len(program.ops))()
    for dst, o in zip(ops, program.ops):
        for name, _ in HelloOp._fields_:
            setattr(dst, name, getattr(o, name))
    bufs = (HelloBuffer * len(program.buffers))()
    for dst, (dom, fpr) in zip(bufs, program.buffers):
        dst.domain, dst.floats_per_row = dom, fpr
    desc = HelloModelDesc(ABI_VERSION, program.window, program.channels0, program.channels1, program.n_experts, int(program.has_meta),
                          int(program.uses_ref), len(program.buffers), bufs, len(program.ops), ops)
    return desc, (ops, bufs)


class Engine:
    """One compiled model resident on one GPU (one instance per process and device)."""

    def __init__(self, spec: ns.ModelSpec, state, device: int = 0, fused: bool = True, winograd: bool = True,
                 program: Optional["compiler.Program"] = None, arithmetic: Optional[str] = None, lanes_max_sites: Optional[int] = None):
        """``program``: an already compiled (or deliberately edited) program to load instead of compiling.
        ``arithmetic``: "fp32" (exact fp32 everywhere, the default), "bf16x3" -- the read convolver's seven 64 -> 64 trunk
        convolutions on the bf16 matrix cores as 3-term splits (x w ~= xh wh + xh wl + xl wh; ~2^-17 per product, the
        residual stream kept in fp32) -- or "bf16x3+32" -- its six 32 -> 32 convolutions too; DESIGN.md section 3.2 gives
        their measured accuracy and speed.  An explicit "bf16x3"
        raises where the mode does not exist (other read-convolver geometries, layer-by-layer paths).  Nothing in the
        environment changes the arithmetic: ``self.program.arithmetic`` is what was asked for here (the test suite runs
        itself in a split mode through a fixture of tests/conftest.py, not through this constructor)."""
        self.lib = load_library()
        self.spec = spec
        self.lanes_max_sites = lanes_max_sites      # None: the module's LANES_MAX_SITES at call time
        # the laned program for small launches is compiled from the same options on first use (None: this engine was handed a
        # finished program, or the model is a single chain)
        self._lanes_recipe = None if program is not None else dict(state=state, fused=fused, winograd=winograd, arithmetic=arithmetic or "fp32")
        if program is None:
            program = compiler.compile_model(spec, state, fused=fused, winograd=winograd, arithmetic=arithmetic or "fp32")
        self.program = program
        p = self.program
        self.device = device
        self.handle, self._desc_arrays = self._create(p)
        self.lanes_handle = None            # a second native engine holding the laned program (multi-chain models, small launches)
        self.lanes_program = None
        self._sequential_only = 0           # > 0 while profiling / debug capture / stamps are armed: they need the sequential program
        self.n_experts = p.n_experts
        self.has_meta = p.has_meta
        self._own = None                    # torch view of the engine's own stream (created on first use)

    def _create(self, program):
        desc, keep = model_desc(program)
        blob = np.ascontiguousarray(program.weights, dtype=np.float32)
        handle = C.c_void_p()
        _check(self.lib.hello_engine_create(C.byref(desc), blob.ctypes.data, blob.nbytes, self.device, C.byref(handle)))
        return handle, keep

    def small_launch_handle(self):
        """The native engine a launch of a few sites should run on: for a model of several independent chains (two read
        technologies, three experts, a meta network: MixtureOfExpertsAdvanced.py:161-252) a second engine holding the LANED
        program -- the same ops and weights, every chain on its own stream (``compiler.assign_lanes``), no buffer shared between
        values -- built on first use; for a single-chain model, or while profiling / debug capture is armed, the engine itself.
        A launch of a few sites is latency-bound: the chains' kernels are a handful of workgroups each and overlap freely."""
        if self._sequential_only or self._lanes_recipe is None:
            return self.handle
        if self.lanes_handle is None:
            r = self._lanes_recipe
            prog = compiler.compile_model(self.spec, r["state"], fused=r["fused"], winograd=r["winograd"], arithmetic=r["arithmetic"], lanes=True)
            if prog.n_lanes == 1:
                self._lanes_recipe = None
                return self.handle
            self.lanes_handle, self._lanes_desc_arrays = self._create(prog)
            self.lanes_program = prog
            self._lanes_recipe = dict(r, state=None)          # (the state is not needed again)
        return self.lanes_handle

    # -- stream ordering of device-path calls ---------------------------------------------------------
    class _OnStream:
        """Device-path calls run on the caller's current torch stream.  When that is the legacy default stream
        (handle 0 -- which the C ABI reads as "the engine's own, non-blocking stream") the call runs on the engine's
        stream bracketed by events: it waits for what the default stream has enqueued (the inputs) and the default
        stream waits for it (the outputs), so the call stays asynchronous and correctly ordered."""

        def __init__(self, engine, device, stream):
            import torch
            self.current = None
            if not stream:                 # None, or 0 = the legacy default stream's handle
                cur = torch.cuda.current_stream(device)
                if cur.cuda_stream == 0:
                    if engine._own is None:
                        engine._own = torch.cuda.ExternalStream(engine.lib.hello_engine_stream(engine.handle), device=device)
                    self.current, self.own = cur, engine._own
                    stream = engine._own.cuda_stream
                else:
                    stream = cur.cuda_stream
            self.handle = stream

        def __enter__(self):
            if self.current is not None:
                self.own.wait_stream(self.current)
            return self.handle

        def __exit__(self, *exc):
            if self.current is not None:
                self.current.wait_stream(self.own)
            return False

    def on_stream(self, device, stream=None):
        return Engine._OnStream(self, device, stream)

    def close(self):
        if getattr(self, "lanes_handle", None):
            self.lib.hello_engine_destroy(self.lanes_handle)
            self.lanes_handle = None
        if getattr(self, "handle", None):
            self.lib.hello_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- batched operator --------------------------------------------------------------------
    def forward(self, reads0, reads_per_allele0, alleles_per_site, reads1=None, reads_per_allele1=None,
                ref_onehot=None, stream: Optional[int] = None, layout_rcl: bool = False,
                out: Optional[Tuple] = None, posteriors: bool = False):
        """reads*: uint8 [R, L, C] NumPy arrays (host path) or torch CUDA tensors (device path; the
        results are then torch CUDA tensors and the call is asynchronous on ``stream``).
        Returns (logits [n_experts, A] float32, meta [S, 3] | None), plus the pair posteriors
        [4, P] as a third element when ``posteriors=True``.  ``out`` = preallocated (logits, meta[,
        posteriors]) device tensors."""
        rpa0 = _i32_host(reads_per_allele0)
        aps = _i32_host(alleles_per_site)
        rpa1 = _i32_host(reads_per_allele1) if reads_per_allele1 is not None else None
        S, A = int(aps.shape[0]), int(rpa0.shape[0])
        on_device = _is_torch(reads0) and reads0.is_cuda
        flags = HELLO_LAYOUT_RCL if layout_rcl else 0
        prog = self.program
        if rpa0.ndim != 1 or aps.ndim != 1 or (rpa1 is not None and rpa1.shape != rpa0.shape):
            raise ValueError("count arrays must be one-dimensional (reads_per_allele1 as long as reads_per_allele0)")

        def prep(x, what, trailing):
            """Only the leading dimension crosses the C ABI, which then reads rows * prod(trailing) bytes:
            every shape, dtype, device and contiguity assumption is checked here (the reference raises a
            Conv1d channel error for a pileup of the wrong width, NNTools.py:633-657)."""
            if x is None:
                return None, 0, None
            if tuple(x.shape[1:]) != trailing or len(x.shape) != 3:
                raise ValueError(f"{what}: expected [rows, {trailing[0]}, {trailing[1]}] "
                                 f"({'[R, C, L]' if layout_rcl and what != 'ref_onehot' else 'channels last'}), "
                                 f"got {tuple(x.shape)}")
            if on_device:
                import torch
                if not (_is_torch(x) and x.is_cuda):
                    raise ValueError(f"{what}: device and host inputs cannot be mixed in one call")
                if x.dtype != torch.uint8:
                    raise TypeError("pileup tensors must be uint8")
                if x.device.index != self.device:
                    raise ValueError(f"{what} lives on cuda:{x.device.index}, the engine on cuda:{self.device}")
                if not x.is_contiguous():
                    raise ValueError(f"{what} must be contiguous")
                return x, x.data_ptr(), x
            if _is_torch(x) and x.is_cuda:
                raise ValueError(f"{what}: device and host inputs cannot be mixed in one call")
            arr = x.numpy() if _is_torch(x) else np.asarray(x)
            if arr.dtype != np.uint8:
                raise TypeError("pileup tensors must be uint8")
            arr = np.ascontiguousarray(arr)
            return arr, arr.ctypes.data, arr

        def trailing(channels):
            return (channels, prog.window) if layout_rcl else (prog.window, channels)

        if prog.channels1 and reads1 is None:
            raise ValueError("this model scores two read technologies: reads1 / reads_per_allele1 are required")
        if not prog.channels1:
            reads1 = rpa1 = None                # like the reference's single-technology forward, which never reads them
        elif rpa1 is None:
            raise ValueError("reads1 given without reads_per_allele1")
        if prog.uses_ref and ref_onehot is None:
            raise ValueError("this model reads the one-hot reference segment: ref_onehot [S, window, 5] is required")
        if not prog.uses_ref:
            ref_onehot = None
        r0, p0, keep0 = prep(reads0, "reads0", trailing(prog.channels0))
        r1, p1, keep1 = prep(reads1, "reads1", trailing(prog.channels1))
        rf, pf, keepf = prep(ref_onehot, "ref_onehot", (prog.window, 5))
        n0 = int(r0.shape[0])
        n1 = int(r1.shape[0]) if r1 is not None else 0
        if rf is not None and int(rf.shape[0]) != S:
            raise ValueError(f"ref_onehot holds {int(rf.shape[0])} segments for {S} sites")
        P = n_pairs(aps) if posteriors else 0
        if on_device:
            import torch
            flags |= HELLO_IN_DEVICE | HELLO_OUT_DEVICE
            post = None
            if out is not None:
                logits, meta = out[0], out[1]
                post = out[2] if len(out) > 2 else None
                for t, need, what in ((logits, self.n_experts * A, "out[0] (logits)"),
                                      (meta if self.has_meta else None, S * 3, "out[1] (meta)"),
                                      (post if posteriors else None, 4 * P, "out[2] (posteriors)")):
                    if t is None:
                        continue
                    if not (_is_torch(t) and t.is_cuda and t.device.index == self.device and t.dtype == torch.float32):
                        raise ValueError(f"{what} must be a float32 tensor on cuda:{self.device}")
                    if not t.is_contiguous() or t.numel() != need:
                        raise ValueError(f"{what} must be contiguous with exactly {need} elements "
                                         f"(got {'a non-contiguous view' if not t.is_contiguous() else t.numel()})")
                if self.has_meta and meta is None:
                    raise ValueError("this model produces meta weights: out[1] is required")
            else:
                logits = torch.empty((self.n_experts, A), dtype=torch.float32, device=reads0.device)
                meta = torch.empty((S, 3), dtype=torch.float32, device=reads0.device) if self.has_meta else None
            if posteriors and post is None:
                post = torch.empty((4, P), dtype=torch.float32, device=reads0.device)
            lp = logits.data_ptr()
            mp = meta.data_ptr() if meta is not None else None
            pp = post.data_ptr() if posteriors else None
        else:
            if out is not None:
                raise ValueError("preallocated outputs are a device-path feature")
            logits = np.empty((self.n_experts, A), dtype=np.float32)
            meta = np.empty((S, 3), dtype=np.float32) if self.has_meta else None
            post = np.empty((4, P), dtype=np.float32) if posteriors else None
            lp = logits.ctypes.data
            mp = meta.ctypes.data if meta is not None else None
            pp = post.ctypes.data if posteriors else None
        # a small launch of a multi-chain model runs the laned program (the chains concurrently; the same bits)
        native = self.small_launch_handle() if S <= (LANES_MAX_SITES if self.lanes_max_sites is None else self.lanes_max_sites) else self.handle
        self._last_native = native

        def launch(handle):
            _check(self.lib.hello_engine_forward(
                native, p0, rpa0.ctypes.data, p1 or None, rpa1.ctypes.data if rpa1 is not None else None,
                aps.ctypes.data, pf or None, S, A, n0, n1, lp, mp, pp, flags, handle))
        if on_device:
            with self.on_stream(reads0.device, stream) as handle:
                launch(handle)
        else:
            launch(stream)
        if posteriors:
            return logits, meta, post
        return logits, meta

    def posteriors(self, logits, meta, alleles_per_site, stream: Optional[int] = None):
        """Genotype-pair posteriors [4, P] (rows mix, expert0, expert1, expert2) for the logits/meta
        of a forward; pairs per site in first-seen itertools.product order."""
        aps = _i32_host(alleles_per_site)
        S, A = int(aps.shape[0]), int(logits.shape[1])
        P = n_pairs(aps)
        on_device = _is_torch(logits) and logits.is_cuda
        if on_device:
            import torch
            out = torch.empty((4, P), dtype=torch.float32, device=logits.device)
            flags = HELLO_IN_DEVICE | HELLO_OUT_DEVICE
            lp, mp, op = logits.data_ptr(), (meta.data_ptr() if meta is not None else None), out.data_ptr()
        else:
            logits = np.ascontiguousarray(logits, dtype=np.float32)
            meta = np.ascontiguousarray(meta, dtype=np.float32) if meta is not None else None
            out = np.empty((4, P), dtype=np.float32)
            flags = 0
            lp, mp, op = logits.ctypes.data, (meta.ctypes.data if meta is not None else None), out.ctypes.data
        if on_device:
            with self.on_stream(logits.device, stream) as handle:
                _check(self.lib.hello_engine_posteriors(self.handle, lp, mp, aps.ctypes.data, S, A, P, op, flags, handle))
        else:
            _check(self.lib.hello_engine_posteriors(self.handle, lp, mp, aps.ctypes.data, S, A, P, op, flags, stream))
        return out

    def synchronize(self):
        _check(self.lib.hello_engine_synchronize(self.handle))
        if self.lanes_handle:
            _check(self.lib.hello_engine_synchronize(self.lanes_handle))

    def last_forward_ms(self) -> float:
        ms = C.c_float()
        _check(self.lib.hello_engine_last_forward_ms(getattr(self, "_last_native", None) or self.handle, C.byref(ms)))
        return float(ms.value)

    def set_profiling(self, max_forwards: int, only: Optional[str] = None):
        """Arm per-op HIP-event timing for the next ``max_forwards`` forwards (0 disarms).  ``only`` = an op
        kind name of ``compiler.OP_NAMES`` ("readconv_fused", ...): record around ops of that kind alone (two
        events per op and forward -- cheap enough for a timed region)."""
        kind = 0 if only is None else {v: k for k, v in compiler.OP_NAMES.items()}[only]
        _check(self.lib.hello_engine_set_profiling_filter(self.handle, kind))
        _check(self.lib.hello_engine_set_profiling(self.handle, int(max_forwards)))
        self._sequential_only = (self._sequential_only & ~1) | (1 if max_forwards > 0 else 0)     # per-op events time the sequential program

    def capture_op_output(self, op_index: Optional[int]):
        """Debug: snapshot the output of op ``op_index`` of every following forward (None disarms)."""
        _check(self.lib.hello_engine_debug_capture(self.handle, -1 if op_index is None else int(op_index)))
        self._sequential_only = (self._sequential_only & ~2) | (0 if op_index is None else 2)

    def read_op_output(self) -> np.ndarray:
        """The snapshot of the last forward, float32 [rows, positions, channels] (rows of the op's domain)."""
        n = C.c_int64()
        _check(self.lib.hello_engine_debug_read(self.handle, None, 0, C.byref(n)))
        out = np.empty(int(n.value), dtype=np.float32)
        _check(self.lib.hello_engine_debug_read(self.handle, out.ctypes.data, out.size, C.byref(n)))
        return out

    def record_stamps(self, mode: int):
        """Diagnostic: 1 = following forwards launch the STAMPED instantiation of the fused read convolver (per-wave s_memtime
        records around every barrier; same results, ~10 % slower), 3 = the same with one workgroup per CU, 0 = off."""
        if not hasattr(self.lib, "hello_engine_debug_stamps"):
            raise RuntimeError(f"{_LIB_PATH} was built before the kernel-timeline diagnostic (hello_engine_debug_stamps): rebuild it")
        _check(self.lib.hello_engine_debug_stamps(self.handle, int(mode)))
        self._sequential_only = (self._sequential_only & ~4) | (4 if mode else 0)

    def read_stamps(self):
        """-> (uint64 [workgroups, waves, groups per workgroup, slots], number of workgroups of the bulk launch) of the last
        stamped forward; slots as documented at hello_engine_debug_stamps in include/hello_mi355x.h."""
        n, layout = C.c_int64(), (C.c_int32 * 5)()
        _check(self.lib.hello_engine_debug_read_stamps(self.handle, None, 0, C.byref(n), layout))
        out = np.empty(int(n.value), dtype=np.uint64)
        _check(self.lib.hello_engine_debug_read_stamps(self.handle, out.ctypes.data, out.size, C.byref(n), layout))
        return out.reshape(layout[0], layout[1], layout[2], layout[3]), int(layout[4])

    def op_times_ms(self):
        """-> (list of (op kind, layer name, mean ms per forward), number of forwards averaged)."""
        n = len(self.program.ops)
        buf = (C.c_float * n)()
        got, nf = C.c_int32(), C.c_int32()
        _check(self.lib.hello_engine_op_times_ms(self.handle, buf, n, C.byref(got), C.byref(nf)))
        rows = [(compiler.OP_NAMES[o.kind], o.name, float(buf[i]) / max(nf.value, 1))
                for i, o in enumerate(self.program.ops[:got.value])]
        return rows, int(nf.value)

    def forward_batch(self, batch, **kw):
        """Convenience over a hello_amd.synth.SiteBatch."""
        return self.forward(batch.reads0, batch.reads_per_allele0, batch.alleles_per_site, batch.reads1,
                            batch.reads_per_allele1, batch.ref_onehot if self.program.uses_ref else None, **kw)
