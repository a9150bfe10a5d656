"""Multi-GPU scoring: sites are independent (no cross-site term anywhere in MoEAttention.forward,
reference python/MixtureOfExpertsAdvanced.py:161-252), so a batch is cut into contiguous site ranges
balanced by read count, every rank scores its range with its own engine (weights replicated, like the
reference's one-model-per-worker process pool, python/call.py:111,215-221), and the per-allele logits
are collected with ONE gather to rank 0 (RCCL over xGMI with the "nccl" backend; "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def partition_sites(reads_per_site: Sequence[int], n_parts: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) site ranges whose read totals are as even as a prefix-sum split allows
    (reads dominate the cost: ~10.15 MFLOP per read vs ~31 MFLOP per allele).  Ranges may be empty when
    there are fewer sites than parts."""
    reads = np.asarray(reads_per_site, dtype=np.int64)
    n = reads.shape[0]
    cum = np.concatenate([[0], np.cumsum(reads)])
    total = cum[-1]
    cuts = [0]
    for k in range(1, n_parts):
        target = total * k / n_parts
        idx = int(np.searchsorted(cum, target, side="left"))
        # choose the nearer of the two neighbouring boundaries
        if idx > 0 and idx <= n and abs(cum[idx - 1] - target) <= abs(cum[min(idx, n)] - target):
            idx -= 1
        cuts.append(min(max(idx, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(n_parts)]


def reads_per_site(batch) -> np.ndarray:
    """Read count of every site of a SiteBatch (both technologies)."""
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    r = np.add.reduceat(batch.reads_per_allele0.astype(np.int64), aoff[:-1])
    if batch.reads_per_allele1 is not None:
        r = r + np.add.reduceat(batch.reads_per_allele1.astype(np.int64), aoff[:-1])
    return r


def gather_rows(local, dst: int = 0, group=None):
    """Gather variable-length [E, n_local] float32 tensors to ``dst`` and concatenate along dim 1 in rank
    order.  One size exchange + one padded gather; returns the concatenation on ``dst``, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([local.shape[1]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    width = max(max(sizes), 1)
    padded = torch.zeros((local.shape[0], width), dtype=local.dtype, device=local.device)
    padded[:, :local.shape[1]] = local
    bucket = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:, :n] for b, n in zip(bucket, sizes)], dim=1)


def score_sharded(score_fn, batch, rank: int, world: int, dst: int = 0, group=None, device=None):
    """Score ``batch`` across ``world`` ranks.  ``score_fn(sub_batch) -> (logits [E, A_sub], meta | None)``
    (NumPy or torch).  Returns (logits [E, A], meta [S, 3] | None) on ``dst``; (None, None) elsewhere."""
    import torch
    lo, hi = partition_sites(reads_per_site(batch), world)[rank]
    n_exp = None
    if hi > lo:
        logits, meta = score_fn(batch.site_slice(lo, hi))
        logits = torch.as_tensor(logits)
        meta = None if meta is None else torch.as_tensor(meta)
        n_exp = logits.shape[0]
    else:
        logits, meta = None, None
    # ranks with an empty range still take part in the collective
    import torch.distributed as dist
    info = torch.tensor([n_exp or 0, 0 if meta is None else 1], dtype=torch.int64)
    if device is not None:
        info = info.to(device)
    dist.all_reduce(info, op=dist.ReduceOp.MAX, group=group)
    n_exp, has_meta = int(info[0].item()), bool(info[1].item())
    dev = device if device is not None else (logits.device if logits is not None else "cpu")
    if logits is None:
        logits = torch.zeros((n_exp, 0), dtype=torch.float32, device=dev)
    logits = logits.to(dev, dtype=torch.float32)
    out = gather_rows(logits, dst, group)
    out_meta = None
    if has_meta:
        m = torch.zeros((3, 0), dtype=torch.float32, device=dev) if meta is None else meta.to(dev).t().contiguous()
        g = gather_rows(m, dst, group)
        out_meta = None if g is None else g.t().contiguous()
    return out, out_meta
