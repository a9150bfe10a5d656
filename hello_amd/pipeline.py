"""Host-resident batches through the engine at device speed (SURVEY.md 8d metric (1), 8e "expected scaling
limiter is host->device input feed").

The caller of the reference holds pileups in host memory (``caller_calling.py:631-639`` builds them from the
featurizer's NumPy arrays).  ``Engine.forward`` on host arrays copies, computes and copies back in sequence;
this module overlaps the three: a batch's pileups cross PCIe on a copy stream into one of ``depth`` device
slots while the previous batch is scored on the compute stream, and results return through pinned buffers.
Only torch's memory/stream plumbing is used; all arithmetic is the engine's.

    pipe = HostPipeline(engine, depth=2)
    for batch in batches:                       # hello_amd.synth.SiteBatch-like: reads0, counts, ...
        for tag, logits, meta, post in pipe.submit(batch, tag=...):
            ...
    for tag, logits, meta, post in pipe.flush():
        ...

Small batches (a few hundred sites) cannot fill the chip on their own: the allele-stage kernels of a 256-site
batch occupy a fraction of the CUs.  Pass several engines of the same model (``engines=[...]``, each with its
own scratch and compute stream) and consecutive batches run concurrently: 260 k -> 360 k sites/s at 256 sites
per batch with four engines, 356 k -> 400 k at 1 024 (at 8 192 sites one engine already fills the GPU).

Results come back in submission order.  A producer that can write straight into pinned memory (the GPU box's
featurizer output, a memory-mapped shard) should pass pinned ``torch.uint8`` tensors: pageable NumPy arrays
are first copied into the slot's pinned staging buffer by the CPU, which costs about as much as scoring them.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .engine import Engine, n_pairs


class _Slot:
    def __init__(self):
        self.pinned_in = [None, None, None]     # reads0, reads1, ref one-hot staging (pinned uint8)
        self.dev_in = [None, None, None]
        self.dev_out = [None, None, None]       # logits, meta, posteriors
        self.pinned_out = [None, None, None]
        self.copied = None
        self.done = None
        self.pending = None                     # (tag, A, S, P) of the batch in flight


def _grow(t, n, **kw):
    import torch
    if t is None or t.numel() < n:
        return torch.empty(int(n * 1.25) + 64, **kw)
    return t


def pin_batch(batch):
    """A SiteBatch whose pileup arrays are pinned torch tensors (slices along the read axis stay pinned and
    contiguous, so ``site_slice`` views of it cross PCIe without a staging copy)."""
    import torch
    from .synth import SiteBatch

    def pin(x):
        return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).pin_memory()
    return SiteBatch(pin(batch.reads0), batch.reads_per_allele0, batch.alleles_per_site, pin(batch.ref_onehot),
                     pin(batch.reads1), batch.reads_per_allele1)


class HostPipeline:
    def __init__(self, engine: Optional[Engine] = None, depth: int = 2, posteriors: bool = True,
                 engines: Optional[List[Engine]] = None):
        import torch
        self.engines = list(engines) if engines else [engine]
        if not self.engines or self.engines[0] is None:
            raise ValueError("an engine is required")
        engine = self.engines[0]
        if any((e.device, e.n_experts, e.has_meta, e.program.window) !=
               (engine.device, engine.n_experts, engine.has_meta, engine.program.window) for e in self.engines):
            raise ValueError("all engines of a pipeline must hold the same model on the same device")
        if depth < 2:
            raise ValueError("depth >= 2 is needed to overlap the copy of one batch with the scoring of another")
        n = len(self.engines)
        depth = -(-max(depth, 2 * n if n > 1 else depth) // n) * n      # a slot always maps to the same engine
        self.engine = engine
        self.posteriors = posteriors
        self.device = torch.device(f"cuda:{engine.device}")
        self.computes = [torch.cuda.Stream(self.device) for _ in self.engines]
        self.compute = self.computes[0]
        self.copy = torch.cuda.Stream(self.device)
        self.slots = [_Slot() for _ in range(depth)]
        for s in self.slots:
            s.copied = torch.cuda.Event()
            s.done = torch.cuda.Event()
        self.count = 0

    # ------------------------------------------------------------------------------------------
    def _stage(self, slot: _Slot, i: int, x):
        """-> pinned uint8 tensor holding ``x`` (flat), ready for an asynchronous copy."""
        import torch
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            if x.dtype != torch.uint8 or x.is_cuda:
                raise TypeError("pileup tensors must be host uint8")
            if x.is_pinned() and x.is_contiguous():
                return x.view(-1)
            x = x.contiguous().numpy()
        x = np.ascontiguousarray(x)
        if x.dtype != np.uint8:
            raise TypeError("pileup tensors must be uint8")
        n = x.size
        slot.pinned_in[i] = _grow(slot.pinned_in[i], n, dtype=torch.uint8, pin_memory=True)
        slot.pinned_in[i][:n].numpy()[...] = x.reshape(-1)
        return slot.pinned_in[i][:n]

    def _harvest(self, slot: _Slot):
        tag, A, S, P = slot.pending
        slot.pending = None
        slot.done.synchronize()
        e = self.engine
        logits = slot.pinned_out[0][:e.n_experts * A].numpy().reshape(e.n_experts, A).copy()
        meta = slot.pinned_out[1][:S * 3].numpy().reshape(S, 3).copy() if e.has_meta else None
        post = slot.pinned_out[2][:4 * P].numpy().reshape(4, P).copy() if self.posteriors else None
        return tag, logits, meta, post

    # ------------------------------------------------------------------------------------------
    def submit(self, batch, tag=None, sink=None) -> List[Tuple]:
        """Queue one batch; returns the batches that finished meanwhile (possibly none), oldest first.
        ``sink`` = (logits_dev [E, >= col + A], meta_dev [>= row + S, 3] | None, col, row): the batch's logits
        (and meta) are also copied, on the compute stream, into these device tensors at that allele column /
        site row -- the multi-GPU path keeps a rank's results resident for the one gather at the end."""
        import torch
        index = self.count % len(self.slots)
        e = self.engines[index % len(self.engines)]
        compute = self.computes[index % len(self.engines)]
        slot = self.slots[index]
        self.count += 1
        finished = [self._harvest(slot)] if slot.pending is not None else []

        window = e.program.window
        shapes = [(-1, window, e.program.channels0), (-1, window, max(e.program.channels1, 1)), (-1, window, 5)]
        hosts = [batch.reads0, batch.reads1 if e.program.channels1 else None,
                 batch.ref_onehot if e.program.uses_ref else None]
        staged = [self._stage(slot, i, h) for i, h in enumerate(hosts)]
        aps = np.ascontiguousarray(batch.alleles_per_site, dtype=np.int32)
        S, A, P = int(aps.shape[0]), int(np.asarray(batch.reads_per_allele0).shape[0]), n_pairs(aps)

        with torch.cuda.stream(self.copy):
            for i, st in enumerate(staged):
                if st is None:
                    continue
                slot.dev_in[i] = _grow(slot.dev_in[i], st.numel(), dtype=torch.uint8, device=self.device)
                slot.dev_in[i][:st.numel()].copy_(st, non_blocking=True)
            slot.copied.record(self.copy)

        f32 = dict(dtype=torch.float32)
        sizes = [e.n_experts * A, S * 3 if e.has_meta else 0, 4 * P if self.posteriors else 0]
        for i, n in enumerate(sizes):
            if n:
                slot.dev_out[i] = _grow(slot.dev_out[i], n, device=self.device, **f32)
                slot.pinned_out[i] = _grow(slot.pinned_out[i], n, pin_memory=True, **f32)
        with torch.cuda.stream(compute):
            compute.wait_event(slot.copied)
            dev = [None if st is None else slot.dev_in[i][:st.numel()].view(shapes[i]) for i, st in enumerate(staged)]
            out = (slot.dev_out[0][:sizes[0]].view(e.n_experts, A),
                   slot.dev_out[1][:sizes[1]].view(S, 3) if sizes[1] else None,
                   slot.dev_out[2][:sizes[2]].view(4, P) if sizes[2] else None)
            e.forward(dev[0], batch.reads_per_allele0, aps, dev[1],
                      batch.reads_per_allele1 if dev[1] is not None else None, dev[2],
                      stream=compute.cuda_stream, out=out, posteriors=self.posteriors)
            if sink is not None:
                sink_logits, sink_meta, col, row = sink
                sink_logits[:, col:col + A].copy_(out[0], non_blocking=True)
                if sink_meta is not None and out[1] is not None:
                    sink_meta[row:row + S].copy_(out[1], non_blocking=True)
            for i, n in enumerate(sizes):
                if n:
                    slot.pinned_out[i][:n].copy_(slot.dev_out[i][:n], non_blocking=True)
            slot.done.record(compute)
        slot.pending = (tag, A, S, P)
        return finished

    def flush(self) -> List[Tuple]:
        """Wait for everything in flight; returns the remaining results in submission order."""
        n = len(self.slots)
        order = [self.slots[(self.count + k) % n] for k in range(n)]      # oldest first
        return [self._harvest(s) for s in order if s.pending is not None]
