"""Multi-GPU scoring: sites are independent (no cross-site term anywhere in MoEAttention.forward,
reference python/MixtureOfExpertsAdvanced.py:161-252), so a batch is cut into contiguous site ranges
balanced by read count, every rank scores its range with its own engine (weights replicated, like the
reference's one-model-per-worker process pool, python/call.py:111,215-221), and the per-allele logits
(+ per-site meta weights of ensemble models, packed behind them in the same buffer) are collected with
exactly ONE collective: a gather to rank 0 (RCCL over xGMI with the "nccl" backend; "gloo" in CPU tests).

Every rank holds the batch's counts, so every rank computes the same partition and therefore every
rank's allele / site counts locally: there is no size exchange and no host-blocking ``.item()``.
"""
from __future__ import annotations

import os
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


def reads_per_site_counts(alleles_per_site, reads_per_allele0, reads_per_allele1=None) -> np.ndarray:
    """Read count of every site from the CSR counts (both technologies)."""
    aps = np.asarray(alleles_per_site, dtype=np.int64)
    aoff = np.concatenate([[0], np.cumsum(aps)])
    r = np.add.reduceat(np.asarray(reads_per_allele0, dtype=np.int64), aoff[:-1])
    if reads_per_allele1 is not None:
        r = r + np.add.reduceat(np.asarray(reads_per_allele1, dtype=np.int64), aoff[:-1])
    return r


def reads_per_site(batch) -> np.ndarray:
    """Read count of every site of a SiteBatch (both technologies)."""
    return reads_per_site_counts(batch.alleles_per_site, batch.reads_per_allele0, batch.reads_per_allele1)


def shard_sizes(alleles_per_site, ranges: Sequence[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """(sites, alleles) of every rank's range -- computed locally by every rank from the shared counts."""
    aoff = np.concatenate([[0], np.cumsum(np.asarray(alleles_per_site, dtype=np.int64))])
    return [(hi - lo, int(aoff[hi] - aoff[lo])) for lo, hi in ranges]


def _row_floats(n_experts: int, has_meta: bool, sites: int, alleles: int) -> int:
    return n_experts * alleles + (3 * sites if has_meta else 0)


def gather_results(logits, meta, sizes: Sequence[Tuple[int, int]], n_experts: int, has_meta: bool,
                   dst: int = 0, group=None):
    """THE collective of the path.  ``logits`` [E, A_rank] and ``meta`` [S_rank, 3] | None (torch tensors on
    the communicator's device; a rank with an empty range passes empty tensors or None) are packed into one
    flat row, padded to the widest rank's row, and gathered to ``dst`` with a single ``dist.gather``.
    ``sizes`` = shard_sizes(...) -- the same list on every rank.  Returns (logits [E, A], meta [S, 3] | None)
    on ``dst`` in rank (= site) order, (None, None) elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if len(sizes) != world:
        raise ValueError(f"{len(sizes)} shard sizes for a world of {world}")
    s_r, a_r = sizes[rank]
    device = _default_device()          # the communicator's: RCCL wants device buffers, gloo host buffers
    width = max(max(_row_floats(n_experts, has_meta, s, a) for s, a in sizes), 1)
    row = torch.zeros(width, dtype=torch.float32, device=device)
    if a_r:
        if tuple(logits.shape) != (n_experts, a_r):
            raise ValueError(f"rank {rank}: logits are {tuple(logits.shape)}, its range holds [{n_experts}, {a_r}]")
        row[:n_experts * a_r] = logits.reshape(-1).to(device=device, dtype=torch.float32)
    if has_meta and s_r:
        if meta is None or tuple(meta.shape) != (s_r, 3):
            raise ValueError(f"rank {rank}: meta is missing or not [{s_r}, 3]")
        row[n_experts * a_r:n_experts * a_r + 3 * s_r] = meta.reshape(-1).to(device=device, dtype=torch.float32)
    bucket = [torch.empty_like(row) for _ in range(world)] if rank == dst else None
    dist.gather(row, bucket, dst=dst, group=group)
    if rank != dst:
        return None, None
    out = torch.cat([b[:n_experts * a].view(n_experts, a) for b, (s, a) in zip(bucket, sizes)], dim=1)
    out_meta = None
    if has_meta:
        out_meta = torch.cat([b[n_experts * a:n_experts * a + 3 * s].view(s, 3) for b, (s, a) in zip(bucket, sizes)], dim=0)
    return out, out_meta


def _default_device():
    """The communicator's device: the current CUDA device under the nccl (= RCCL) backend, else the CPU."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def score_sharded(score_fn, batch, rank: int, world: int, n_experts: int = 1, has_meta: bool = False,
                  dst: int = 0, group=None, device=None):
    """Score ``batch`` across ``world`` ranks.  ``score_fn(sub_batch) -> (logits [E, A_sub], meta | None)``
    (NumPy or torch).  ``n_experts`` / ``has_meta`` are properties of the model (``Engine.n_experts`` /
    ``Engine.has_meta``): a rank whose range is empty still knows the row layout.  Returns (logits [E, A],
    meta [S, 3] | None) on ``dst``; (None, None) elsewhere.  One collective (``gather_results``)."""
    import torch
    ranges = partition_sites(reads_per_site(batch), world)
    sizes = shard_sizes(batch.alleles_per_site, ranges)
    lo, hi = ranges[rank]
    dev = torch.device(device) if device is not None else _default_device()
    logits = meta = None
    if hi > lo:
        logits, meta = score_fn(batch.site_slice(lo, hi))
        logits = torch.as_tensor(logits).to(dev)
        meta = None if meta is None else torch.as_tensor(meta).to(dev)
    else:
        logits = torch.zeros((n_experts, 0), dtype=torch.float32, device=dev)
    return gather_results(logits, meta, sizes, n_experts, has_meta, dst, group)


# ------------------------------------------------------------------------------------------------
# host side of one rank: CPU affinity near its GPU
# ------------------------------------------------------------------------------------------------
def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_numa_node(device_index: int) -> int:
    """NUMA node of a GPU from sysfs, -1 when unknown."""
    try:
        import torch
        prop = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(prop, 'pci_domain_id', 0):04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
        return int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    except Exception:
        return -1


def rank_cpus(local_rank: int, local_world: int, device_index: Optional[int] = None,
              node_of_device=None, cpus_of_node=None, n_devices: Optional[int] = None) -> List[int]:
    """CPUs one rank's host threads (pinned staging copies, CSR building, shard readers, the record stage) should run
    on.  Local rank r drives GPU r modulo the number of GPUs; when sysfs names the NUMA node of those GPUs, the ranks
    whose GPUs sit on the same node split THAT node's CPUs evenly, by the rank's index among them -- every CPU of a
    node with GPUs goes to exactly one rank.  Otherwise a plain equal split of the affinity mask.  ``node_of_device`` /
    ``cpus_of_node`` / ``n_devices`` replace the sysfs and device queries (tests)."""
    allowed = sorted(os.sched_getaffinity(0))
    pool, index, sharers = allowed, local_rank, max(local_world, 1)
    if device_index is not None:
        try:
            if n_devices is None:
                import torch
                n_devices = max(torch.cuda.device_count(), 1)
            node_of_device = node_of_device or _gpu_numa_node
            if cpus_of_node is None:
                def cpus_of_node(node):
                    return _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
            nodes = [node_of_device(r % n_devices) for r in range(local_world)]
            node = nodes[local_rank]
            cpus = [c for c in cpus_of_node(node) if c in set(allowed)] if node >= 0 else []
            if cpus:
                same = [r for r in range(local_world) if nodes[r] == node]
                pool, index, sharers = cpus, same.index(local_rank), len(same)
        except Exception:
            pool, index, sharers = allowed, local_rank, max(local_world, 1)
    lo, hi = len(pool) * index // sharers, len(pool) * (index + 1) // sharers
    return pool[lo:hi] or pool or allowed


def pin_rank(local_rank: int, local_world: int, device_index: Optional[int] = None) -> List[int]:
    """Apply ``rank_cpus`` to this process (call before allocating the rank's pinned pool, so first touch lands
    near the GPU).  Returns the CPUs set; never raises (a container may forbid it)."""
    cpus = rank_cpus(local_rank, local_world, device_index)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        pass
    return cpus


# ------------------------------------------------------------------------------------------------
# evidence of a multi-rank run: who ran where, on what, for how long (collected AFTER the timed region)
# ------------------------------------------------------------------------------------------------
def device_identity(device_index: int) -> dict:
    """What distinguishes one GPU of the node from another, from the runtime's own properties: PCI address, UUID, name.
    ``identity_source`` says which of them the runtime really provided ("pci_bus_id+uuid", "pci_bus_id", "uuid") or
    "device_index" when neither could be read -- with ``identity_error`` naming why; nothing is swallowed silently."""
    out = {"device_index": int(device_index), "pci_bus_id": None, "uuid": None, "name": None,
           "identity_source": "device_index", "identity_error": None}
    try:
        import torch
        prop = torch.cuda.get_device_properties(device_index)
    except Exception as exc:
        out["identity_error"] = f"get_device_properties: {exc!r}"
        return out
    out["name"] = getattr(prop, "name", None)
    errors = []
    try:
        out["pci_bus_id"] = f"{getattr(prop, 'pci_domain_id', 0):04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
    except Exception as exc:
        errors.append(f"pci: {exc!r}")
    try:
        uuid = str(prop.uuid)
        # a real UUID is 32 hex digits (with or without dashes / a "GPU-" prefix); anything else is not an identity
        digits = uuid.lower().replace("gpu-", "").replace("-", "")
        if len(digits) >= 16 and all(c in "0123456789abcdef" for c in digits) and set(digits) != {"0"}:
            out["uuid"] = uuid
        else:
            errors.append(f"uuid: not a UUID: {uuid!r}")
    except Exception as exc:
        errors.append(f"uuid: {exc!r}")
    src = [k for k in ("pci_bus_id", "uuid") if out[k]]
    out["identity_source"] = "+".join(src) if src else "device_index"
    out["identity_error"] = "; ".join(errors) or None
    return out


def collect_rank_reports(report: dict, group=None) -> List[dict]:
    """Every rank's ``report`` on every rank, in rank order (one ``all_gather_object``; a plain list of the one report when no
    process group is up).  Evidence, not data path: called after the timed region's closing fence."""
    try:
        import torch.distributed as dist
        up = dist.is_available() and dist.is_initialized()
    except Exception:
        up = False
    if not up:
        return [dict(report)]
    bucket = [None] * dist.get_world_size(group)
    dist.all_gather_object(bucket, dict(report), group=group)
    return bucket


def summarize_ranks(reports: Sequence[dict]) -> dict:
    """The audit fields of a multi-rank bench line from the ranks' own reports: the list itself, how many DISTINCT devices the
    ranks drove, which rank was slowest, and how even the read-balanced partition came out (min / max reads over the ranks that
    had any).  A device is keyed on (host, PCI address, UUID) TOGETHER.  When any rank could name neither (its
    ``identity_source`` is "device_index": ranks isolated by HIP_VISIBLE_DEVICES all see index 0, ranks sharing a card see
    different ones), a count would prove nothing: ``distinct_devices`` is then None and ``identity_warning`` says why."""
    reports = sorted((dict(r) for r in reports), key=lambda r: r.get("rank", 0))
    for r in reports:
        r.setdefault("identity_source", "+".join(k for k in ("pci_bus_id", "uuid") if r.get(k)) or "device_index")
    unnamed = [int(r.get("rank", 0)) for r in reports if r["identity_source"] == "device_index"]
    seconds = [float(r.get("timed_seconds", 0.0)) for r in reports]
    reads = [int(r.get("reads", 0)) for r in reports]
    busy = [x for x in reads if x > 0]
    return {
        "ranks": reports, "ranks_seen": len(reports),
        "distinct_devices": None if unnamed else len({(r.get("host"), r.get("pci_bus_id"), r.get("uuid")) for r in reports}),
        "identity_sources": sorted({r["identity_source"] for r in reports}),
        "identity_warning": (f"rank(s) {unnamed} could name their device only by its index: distinct devices cannot be counted"
                             if unnamed else None),
        "slowest_rank": int(reports[int(np.argmax(seconds))].get("rank", 0)) if reports else None,
        "rank_seconds_min_max": [round(min(seconds), 6), round(max(seconds), 6)] if seconds else None,
        "balance": round(min(busy) / max(busy), 4) if busy else None,
    }
