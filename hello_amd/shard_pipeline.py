"""Shards -> records at the engine's rate: the GPU and host stages of ``python -m hello_amd.call``.

The reference runs ``caller_calling.main`` per shard in a process pool, one site at a time (python/call.py:111,215-221;
python/caller_calling.py:859-900).  Here the unit of GPU work is decoupled from the unit of file work:

    reader threads      shard file -> ``PackedShard`` (validated flat arrays, featurizer index arithmetic)
    ``ShardScorer``     SEVERAL shards coalesced into one launch of ~8 k sites: their fifteen featurizer arrays are laid
                        out in ONE pinned block (a single pass over the bytes), cross PCIe in ONE copy on a copy stream
                        into one of ``depth`` device slots, and ``hello_engine_featurize`` -> ``hello_engine_forward`` (+
                        posteriors) run on the compute stream while the next block is being laid out and copied; pair
                        posteriors + meta weights return through pinned memory
    ``RecordWriter``    ``hello_site_records`` (multi-threaded C, hello_amd/records.py) turns a launch's posteriors into
                        every shard's ``.vcf`` lines, ``.features`` pickle stream and final-VCF lines; Python only slices
                        the blobs per shard and writes files -- no per-site Python object anywhere

Only torch's memory / stream plumbing is used; all arithmetic is the engine's.
"""
from __future__ import annotations

import os
import queue
import threading
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import records as rec
from .engine import n_pairs
from .featurizer import FEATURIZE_ARRAYS, featurize_device
from .shards import PackedShard

SENTINEL = "Completed running the script"            # caller_calling.py:902, checked by call.py:225-229
PER_READ = ("ref_start", "mapq", "orientation", "hp", "site_of_read")
SHARED = ("ref", "ref_off", "window_start", "asm_start", "asm_stop")          # the same for both technologies
DTYPES = dict(bases=np.uint8, quals=np.uint8, read_off=np.int64, cigars=np.uint32, cigar_off=np.int64, ref_start=np.int64,
              mapq=np.uint8, orientation=np.int8, hp=np.uint8, site_of_read=np.int32, ref=np.uint8, ref_off=np.int64,
              window_start=np.int64, asm_start=np.int64, asm_stop=np.int64)


def prepare(shard: PackedShard, hybrid: bool, uses_ref: bool) -> PackedShard:
    """Reader-thread work on a loaded shard: everything that does not need the GPU."""
    if shard.n_sites == 0:
        return shard
    if hybrid and not shard.has_reads(1):
        raise ValueError("this model scores two read technologies: every allele of the shard needs both read sets")
    shard.featurizer_core(0)
    if hybrid:
        shard.featurizer_core(1)
    if uses_ref and shard.n_sites:
        shard.onehot = shard.segment_onehot()
    return shard


@dataclass
class Scored:
    """One launch's results on the host."""
    shards: List[PackedShard]
    tags: list
    posteriors: np.ndarray          # float32 [4, P]
    meta: Optional[np.ndarray]      # float32 [S, 3]
    seconds: float = 0.0


class _Slot:
    def __init__(self):
        self.pinned = self.dev = None           # staging block (uint8) on both sides
        self.pile = [None, None]                # featurizer output per technology (device uint8)
        self.out_dev = self.out_pinned = None   # float32: logits | meta | posteriors
        self.copied = self.done = None
        self.pending = None


def _grow(t, n, **kw):
    import torch
    if t is None or t.numel() < n:
        return torch.empty(int(n * 1.25) + 4096, **kw)
    return t


class ShardScorer:
    """Featurise + score coalesced shards on the GPU, ``depth`` launches in flight."""

    def __init__(self, network, include_hp: bool = False, feature_length: int = 150, depth: int = 2):
        import torch
        self.engine = eng = network.engine
        prog = eng.program
        self.hybrid = bool(prog.channels1)
        self.uses_ref = bool(prog.uses_ref)
        want0 = 7 if include_hp else 6
        if prog.channels0 != want0:
            raise ValueError(f"--include_hp {'set' if include_hp else 'not set'}: the featurizer would write {want0} channels, "
                             f"the model reads {prog.channels0}")
        if prog.window != feature_length:
            raise ValueError(f"the model reads {prog.window} bp windows, the driver featurises {feature_length}")
        self.L = feature_length
        self.channels = [prog.channels0, prog.channels1]
        self.device = torch.device(f"cuda:{eng.device}")
        self.compute = torch.cuda.Stream(self.device)
        self.copy = torch.cuda.Stream(self.device)
        self.slots = [_Slot() for _ in range(max(depth, 2))]
        for s in self.slots:
            s.copied, s.done = torch.cuda.Event(), torch.cuda.Event()
        self.count = 0
        self.stage_seconds = 0.0

    # -- layout of one launch's staging block ---------------------------------------------------------------------
    def _layout(self, shards: Sequence[PackedShard]):
        techs = (0, 1) if self.hybrid else (0,)
        S = sum(sh.n_sites for sh in shards)
        parts, at = {}, 0

        def place(key, dtype, count):
            nonlocal at
            parts[key] = (at, np.dtype(dtype), count)
            at += (count * np.dtype(dtype).itemsize + 15) & ~15
        reads = {}
        for t in techs:
            fa = [sh.featurizer_core(t) for sh in shards]
            reads[t] = sum(int(f["site_of_read"].shape[0]) for f in fa)
            place(("bases", t), np.uint8, sum(int(f["bases"].shape[0]) for f in fa) + 1)
            place(("quals", t), np.uint8, sum(int(f["quals"].shape[0]) for f in fa) + 1)
            place(("cigars", t), np.uint32, sum(int(f["cigars"].shape[0]) for f in fa) + 1)
            place(("read_off", t), np.int64, reads[t] + 1)
            place(("cigar_off", t), np.int64, reads[t] + 1)
            for name in PER_READ:
                place((name, t), DTYPES[name], reads[t])
        place(("ref", None), np.uint8, sum(int(sh.ref.shape[0]) for sh in shards) + 1)
        place(("ref_off", None), np.int64, S + 1)
        for name in ("window_start", "asm_start", "asm_stop"):
            place((name, None), np.int64, S)
        if self.uses_ref:
            place(("onehot", None), np.uint8, S * self.L * 5)
        return parts, at, reads, S

    def _fill(self, shards, parts, block: np.ndarray, techs):
        """One pass over the launch's bytes: every shard's arrays land at their place in the pinned block, offsets and
        site indices shifted to the coalesced numbering.  ONE NumPy call per field and launch (a concatenation straight into
        the pinned view, then one vector add of the per-shard shifts), not one per field and shard: with reference-sized shards
        (400 sites, ~20 per launch) the per-shard form spent more host time in Python call overhead than in copying."""
        def view(key):
            at, dtype, count = parts[key]
            return block[at:at + count * dtype.itemsize].view(dtype)

        def cat(key, arrays, tail=None):
            v = view(key)
            n = sum(int(a.shape[0]) for a in arrays)
            np.concatenate(arrays, out=v[:n], casting="same_kind")
            if tail is not None:
                v[n] = tail
            return v, n

        def shifted(v, counts, shifts):
            """v[k-th run of counts[k] elements] += shifts[k] (runs in order; a zero shift everywhere is skipped)."""
            if len(shifts) > 1 and any(shifts):
                v += np.repeat(np.asarray(shifts, dtype=v.dtype), np.asarray(counts, dtype=np.int64))
        n_sites = [sh.n_sites for sh in shards]
        site_shift = np.concatenate([[0], np.cumsum(n_sites)[:-1]]).tolist()
        for t in techs:
            fa = [sh.featurizer_core(t) for sh in shards]
            for name in ("bases", "quals", "cigars"):
                cat((name, t), [f[name] for f in fa], tail=0)
            for name, data in (("read_off", "bases"), ("cigar_off", "cigars")):
                v = view((name, t))
                v[0] = 0
                _, n = cat_into(v[1:], [f[name][1:] for f in fa])
                sizes = [int(f[data].shape[0]) for f in fa]
                shifted(v[1:1 + n], [int(f[name].shape[0]) - 1 for f in fa], np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist())
            for name in PER_READ:
                v, n = cat((name, t), [f[name] for f in fa])
                if name == "site_of_read":
                    shifted(v[:n], [int(f[name].shape[0]) for f in fa], site_shift)
        cat(("ref", None), [sh.ref for sh in shards], tail=0)
        v = view(("ref_off", None))
        v[0] = 0
        _, n = cat_into(v[1:], [sh.ref_off[1:] for sh in shards])
        ref_sizes = [int(sh.ref.shape[0]) for sh in shards]
        shifted(v[1:1 + n], n_sites, np.concatenate([[0], np.cumsum(ref_sizes)[:-1]]).tolist())
        for name, attr in (("window_start", "window_start"), ("asm_start", "start"), ("asm_stop", "stop")):
            cat((name, None), [getattr(sh, attr) for sh in shards])
        if self.uses_ref:
            cat(("onehot", None), [sh.onehot.reshape(-1) for sh in shards if sh.n_sites])

    # -- pipeline ---------------------------------------------------------------------------------------------------
    def _harvest(self, slot: _Slot) -> Scored:
        shards, tags, A, S, P, t0 = slot.pending
        slot.pending = None
        slot.done.synchronize()
        e = self.engine
        n_logits, n_meta = e.n_experts * A, (3 * S if e.has_meta else 0)
        host = slot.out_pinned.numpy()
        meta = host[n_logits:n_logits + n_meta].reshape(S, 3).copy() if e.has_meta else None
        post = host[n_logits + n_meta:n_logits + n_meta + 4 * P].reshape(4, P).copy()
        return Scored(shards, tags, post, meta, time.perf_counter() - t0)

    def submit(self, shards: Sequence[PackedShard], tags: Optional[list] = None) -> List[Scored]:
        """Queue one launch over ``shards`` (coalesced in order); returns what finished meanwhile, oldest first."""
        import torch
        shards = [sh for sh in shards]
        tags = list(tags) if tags is not None else [None] * len(shards)
        slot = self.slots[self.count % len(self.slots)]
        self.count += 1
        finished = [self._harvest(slot)] if slot.pending is not None else []
        live = [(sh, tg) for sh, tg in zip(shards, tags) if sh.n_sites]
        if not live:                                               # nothing to launch: empty shards still get their files
            finished.append(Scored(shards, tags, np.zeros((4, 0), np.float32), None))
            return finished
        t0 = time.perf_counter()
        e = self.engine
        techs = (0, 1) if self.hybrid else (0,)
        scored = [sh for sh, _ in live]
        parts, nbytes, reads, S = self._layout(scored)
        slot.pinned = _grow(slot.pinned, nbytes, dtype=torch.uint8, pin_memory=True)
        self._fill(scored, parts, slot.pinned.numpy(), techs)
        self.stage_seconds += time.perf_counter() - t0
        with torch.cuda.stream(self.copy):
            slot.dev = _grow(slot.dev, nbytes, dtype=torch.uint8, device=self.device)
            slot.dev[:nbytes].copy_(slot.pinned[:nbytes], non_blocking=True)
            slot.copied.record(self.copy)

        rpa = [np.concatenate([sh.featurizer_core(t)["reads_per_allele"] for sh in scored]) for t in techs]
        aps = np.concatenate([sh.alleles_per_site for sh in scored]).astype(np.int32)
        A, P = int(rpa[0].shape[0]), n_pairs(aps)
        sizes = [e.n_experts * A, 3 * S if e.has_meta else 0, 4 * P]
        total = sum(sizes)
        slot.out_dev = _grow(slot.out_dev, total, dtype=torch.float32, device=self.device)
        slot.out_pinned = _grow(slot.out_pinned, total, dtype=torch.float32, pin_memory=True)
        base = slot.dev.data_ptr()
        with torch.cuda.stream(self.compute):
            self.compute.wait_event(slot.copied)
            pile = []
            for t in techs:
                n = reads[t] * self.L * self.channels[t]
                slot.pile[t] = _grow(slot.pile[t], n, dtype=torch.uint8, device=self.device)
                ptr = {name: base + parts[(name, t if (name, t) in parts else None)][0] for name in FEATURIZE_ARRAYS}
                featurize_device(e, ptr, reads[t], S, self.L, self.channels[t], slot.pile[t].data_ptr(), self.compute.cuda_stream)
                pile.append(slot.pile[t][:n].view(reads[t], self.L, self.channels[t]))
            seg = None
            if self.uses_ref:
                at = parts[("onehot", None)][0]
                seg = slot.dev[at:at + S * self.L * 5].view(S, self.L, 5)
            out = (slot.out_dev[:sizes[0]].view(e.n_experts, A),
                   slot.out_dev[sizes[0]:sizes[0] + sizes[1]].view(S, 3) if sizes[1] else None,
                   slot.out_dev[sizes[0] + sizes[1]:total].view(4, P))
            e.forward(pile[0], rpa[0], aps, pile[1] if self.hybrid else None, rpa[1] if self.hybrid else None, seg,
                      stream=self.compute.cuda_stream, out=out, posteriors=True)
            slot.out_pinned[:total].copy_(slot.out_dev[:total], non_blocking=True)
            slot.done.record(self.compute)
        slot.pending = (shards, tags, A, S, P, t0)
        return finished

    def flush(self) -> List[Scored]:
        n = len(self.slots)
        order = [self.slots[(self.count + k) % n] for k in range(n)]      # oldest first
        return [self._harvest(s) for s in order if s.pending is not None]


def cat_into(v: np.ndarray, arrays):
    """Concatenate 1-D ``arrays`` into the head of ``v`` (one C call); -> (v, elements written)."""
    n = sum(int(a.shape[0]) for a in arrays)
    np.concatenate(arrays, out=v[:n], casting="same_kind")
    return v, n


# ---------------------------------------------------------------------------------------------------------------------
# record stage
# ---------------------------------------------------------------------------------------------------------------------
def site_table(shards: Sequence[PackedShard], genomes: Optional[Dict[str, bytes]] = None, wanted=None, keep=None) -> rec.SiteTable:
    """The record stage's view of the sites of coalesced shards: the shards' byte tables concatenated, their chromosome
    indices renumbered into the union of their (few) names -- no per-site or per-allele Python object."""
    live = [sh for sh in shards if sh.n_sites]
    names = sorted({n for sh in live for n in sh.chromosome_names})
    index = {n: i for i, n in enumerate(names)}
    chrom_of = np.concatenate([np.array([index[n] for n in sh.chromosome_names], np.int32)[sh.chromosome_of_site] for sh in live])

    def cat_offsets(offs, sizes):
        shifts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        return np.concatenate([[0]] + [o[1:] + shift for o, shift in zip(offs, shifts)]).astype(np.int64)
    ref = np.concatenate([sh.ref for sh in live] + [np.zeros(1, np.uint8)])
    ref_off = cat_offsets([sh.ref_off for sh in live], [sh.ref.shape[0] for sh in live])
    text = np.concatenate([sh.allele_text for sh in live] + [np.zeros(1, np.uint8)])
    text_off = cat_offsets([sh.allele_text_off for sh in live], [sh.allele_text.shape[0] for sh in live])
    if keep is None and wanted is not None:
        keep = np.array([n in wanted for n in names], np.uint8)[chrom_of]
    return rec.SiteTable(np.concatenate([sh.alleles_per_site for sh in live]), text, text_off, names, chrom_of,
                         np.concatenate([sh.start for sh in live]), np.concatenate([sh.stop for sh in live]),
                         ref, ref_off, np.concatenate([sh.window_start for sh in live]), genomes=genomes, keep=keep)


@dataclass
class ShardOutput:
    """What one shard left on disk, and what the final sort needs of it."""
    tag: object
    prefix: str
    n_sites: int
    n_records: int
    chromosomes: List[str]                 # names indexed by ``chromosome_of``
    chromosome_of: np.ndarray              # per final-VCF line
    position: np.ndarray                   # per final-VCF line (0-based, normalised)
    line_bytes: np.ndarray                 # per final-VCF line


class RecordWriter:
    """Turns ``Scored`` launches into the per-shard files of the reference's caller (``<prefix>.vcf``, ``.features``,
    ``.log`` with the sentinel) plus ``<prefix>.mean.vcf`` (the shard's lines of the final VCF, what prepareVcf.py:126-176
    writes into its temporary directory); the lines' sort keys stay in memory (``outputs``; ``save_index`` writes them
    for another rank to merge)."""

    def __init__(self, prefix_of, genomes: Optional[Dict[str, bytes]] = None, wanted=None, threads: int = 0):
        self.prefix_of, self.genomes, self.wanted, self.threads = prefix_of, genomes, wanted, threads
        self.outputs: List[ShardOutput] = []
        self.seconds = 0.0

    def write(self, scored: Scored):
        t0 = time.perf_counter()
        live = [(sh, tg) for sh, tg in zip(scored.shards, scored.tags) if sh.n_sites]
        if live:
            table = site_table([sh for sh, _ in live], self.genomes, self.wanted)
            site_off = np.concatenate([[0], np.cumsum([sh.n_sites for sh, _ in live])]).astype(np.int32)
            with rec.site_records(table, scored.posteriors, scored.meta, site_off, features=True, threads=self.threads) as r:
                for k, (sh, tag) in enumerate(live):
                    lo, hi = int(site_off[k]), int(site_off[k + 1])
                    prefix = self.prefix_of(tag)
                    with open(prefix + ".vcf", "wb") as fh:
                        fh.write(r.shard_vcf[int(r.shard_vcf_off[lo]):int(r.shard_vcf_off[hi])])
                    with open(prefix + ".features", "wb") as fh:
                        fh.write(r.features[int(r.features_off[k]):int(r.features_off[k + 1])])
                    with open(prefix + ".mean.vcf", "wb") as fh:
                        fh.write(r.mean_vcf[int(r.mean_vcf_off[lo]):int(r.mean_vcf_off[hi])])
                    nbytes = np.diff(r.mean_vcf_off[lo:hi + 1])
                    has = nbytes > 0
                    out = ShardOutput(tag, prefix, sh.n_sites, int(r.n_records[k]), table.names,
                                      table.chromosome_of_site[lo:hi][has].copy(), r.mean_position[lo:hi][has].copy(),
                                      nbytes[has].astype(np.int64))
                    self._log(sh, out, scored.seconds)
                    self.outputs.append(out)
        for sh, tag in zip(scored.shards, scored.tags):
            if not sh.n_sites:                                       # an empty shard still completes (call.py:225-229)
                prefix = self.prefix_of(tag)
                for suffix, payload in ((".vcf", b""), (".mean.vcf", b"")):
                    open(prefix + suffix, "wb").write(payload)
                import pickle
                pickle.dump([], open(prefix + ".features", "wb"))
                out = ShardOutput(tag, prefix, 0, 0, [], np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.int64))
                self._log(sh, out, 0.0)
                self.outputs.append(out)
        self.seconds += time.perf_counter() - t0

    @staticmethod
    def _log(shard: PackedShard, out: ShardOutput, seconds: float):
        with open(out.prefix + ".log", "w") as log:
            log.write(f"Shard {getattr(shard, 'path', '<memory>')}: {shard.n_sites} candidate sites\n")
            log.write("".join("Completed %d sites\n" % n for n in range(100, shard.n_sites + 1, 100)))   # caller_calling.py:888-891
            log.write(f"Scored {shard.n_sites} sites, {out.n_records} records (launch of {seconds:.3f} s)\n")
            log.write(SENTINEL + "\n")


def run(network, shard_paths: Sequence[str], prefix_of, include_hp: bool = False, genomes=None, wanted=None,
        reader_threads: int = 4, record_threads: int = 0, sites_per_launch: int = 8192, reads_per_launch: int = 320_000,
        read_ahead: Optional[int] = None, depth: int = 2, tags: Optional[list] = None, loader=None, scorer=None,
        writer_threads: int = 2) -> "RunStats":
    """Score shard files in order with bounded read-ahead: at most ``read_ahead`` loaded shards wait for the GPU, at most
    ``depth`` launches are in flight and at most two scored launches wait for the record writer, whatever the number of
    shards -- host memory is flat over a run."""
    from concurrent.futures import ThreadPoolExecutor
    scorer = scorer or ShardScorer(network, include_hp, depth=depth)
    writer = RecordWriter(prefix_of, genomes, wanted, record_threads)
    tags = list(tags) if tags is not None else list(range(len(shard_paths)))
    loader = loader or PackedShard.from_file
    read_ahead = read_ahead or max(2 * reader_threads, 4)
    results: "queue.Queue" = queue.Queue(maxsize=2)
    failure: List[BaseException] = []
    position = {tag: i for i, tag in enumerate(tags)}

    def record_loop():
        while True:
            item = results.get()
            if item is None:
                return
            if failure:
                continue                                             # drain so the producer never blocks
            try:
                writer.write(item)
            except BaseException as e:                               # noqa: BLE001 -- re-raised by the caller's thread
                failure.append(e)

    def load(path):
        return prepare(loader(path), scorer.hybrid, scorer.uses_ref)

    stats = RunStats()
    t_start = time.perf_counter()
    # the record stage's C call is multi-threaded; the per-shard file writes around it are not: two writer threads take
    # launches alternately (a launch's shards are written by one thread; the final sort orders records by key, then shard)
    threads = [threading.Thread(target=record_loop, name=f"hello-records-{k}", daemon=True) for k in range(max(1, writer_threads))]
    for thread in threads:
        thread.start()
    try:
        with ThreadPoolExecutor(max_workers=max(1, reader_threads), thread_name_prefix="hello-reader") as pool:
            futures, nxt = [], 0
            batch, batch_tags, sites, reads = [], [], 0, 0

            def launch():
                nonlocal batch, batch_tags, sites, reads
                for done in scorer.submit(batch, batch_tags):
                    results.put(done)
                stats.launches += 1
                batch, batch_tags, sites, reads = [], [], 0, 0

            for i in range(len(shard_paths)):
                while nxt < len(shard_paths) and len(futures) < read_ahead:
                    futures.append(pool.submit(load, shard_paths[nxt]))
                    nxt += 1
                t0 = time.perf_counter()
                shard = futures.pop(0).result()
                stats.wait_read += time.perf_counter() - t0
                if failure:
                    raise failure[0]
                n_reads = shard.n_reads(0) + (shard.n_reads(1) if scorer.hybrid else 0)
                if batch and (sites + shard.n_sites > sites_per_launch or reads + n_reads > reads_per_launch):
                    launch()
                batch.append(shard)
                batch_tags.append(tags[i])
                sites += shard.n_sites
                reads += n_reads
                stats.sites += shard.n_sites
                stats.reads += n_reads
            if batch:
                launch()
            for done in scorer.flush():
                results.put(done)
    finally:
        for _ in threads:
            results.put(None)
        for thread in threads:
            thread.join()
    if failure:
        raise failure[0]
    stats.seconds = time.perf_counter() - t_start
    stats.stage_seconds, stats.record_seconds = scorer.stage_seconds, writer.seconds
    stats.outputs = sorted(writer.outputs, key=lambda o: position.get(o.tag, 0))       # shard order, whatever thread wrote them
    return stats


@dataclass
class RunStats:
    sites: int = 0
    reads: int = 0
    launches: int = 0
    seconds: float = 0.0
    wait_read: float = 0.0          # the GPU feeder waiting for a reader thread
    stage_seconds: float = 0.0      # laying launches out in pinned memory
    record_seconds: float = 0.0     # record stage (its own thread)
    outputs: Optional[List[ShardOutput]] = None


# ---------------------------------------------------------------------------------------------------------------------
# final VCF: the per-shard mean lines, sorted (prepareVcf.py:199-260; ``vcf-sort`` replaced by an in-process sort)
# ---------------------------------------------------------------------------------------------------------------------
def save_index(path: str, outputs: Sequence[ShardOutput]) -> str:
    """One rank's sort keys, for the rank that writes the final VCF."""
    names = sorted({c for o in outputs for c in o.chromosomes})
    rank_of = {c: i for i, c in enumerate(names)}
    cat = lambda xs, dtype: np.concatenate([np.asarray(x, dtype) for x in xs] + [np.zeros(0, dtype)])      # noqa: E731
    np.savez(path, prefixes=np.array([o.prefix for o in outputs] or [""]), n_shards=np.array(len(outputs)),
             lines=np.array([o.position.shape[0] for o in outputs], np.int64), chromosomes=np.array(names or [""]),
             chromosome_of=cat([np.array([rank_of[c] for c in o.chromosomes], np.int32)[o.chromosome_of] if o.position.size
                                else np.zeros(0, np.int32) for o in outputs], np.int32),
             position=cat([o.position for o in outputs], np.int64), line_bytes=cat([o.line_bytes for o in outputs], np.int64))
    return path


def load_index(path: str) -> List[ShardOutput]:
    with np.load(path, allow_pickle=False) as z:
        n = int(z["n_shards"])
        names = [str(c) for c in z["chromosomes"]]
        cuts = np.concatenate([[0], np.cumsum(z["lines"][:n])]).astype(np.int64)
        return [ShardOutput(None, str(z["prefixes"][k]), -1, int(cuts[k + 1] - cuts[k]), names,
                            z["chromosome_of"][cuts[k]:cuts[k + 1]], z["position"][cuts[k]:cuts[k + 1]],
                            z["line_bytes"][cuts[k]:cuts[k + 1]]) for k in range(n)]


def merge_final_vcf(outputs: Sequence[ShardOutput], header_of, output_path: str) -> int:
    """Write ``output_path``: header + every shard's final-VCF lines sorted by (chromosome name, position), ties in
    shard then site order.  Only the keys of all records are held at once (20 bytes per record); the lines are moved a
    shard at a time from its ``.mean.vcf`` into their place in a memory-mapped output.  -> number of records."""
    names = sorted({c for o in outputs for c in (o.chromosomes[i] for i in np.unique(o.chromosome_of))})
    rank_of = {c: i for i, c in enumerate(names)}
    chrom = np.concatenate([np.array([rank_of.get(c, -1) for c in o.chromosomes], np.int64)[o.chromosome_of] if o.position.size
                            else np.zeros(0, np.int64) for o in outputs] + [np.zeros(0, np.int64)])
    pos = np.concatenate([o.position for o in outputs] + [np.zeros(0, np.int64)])
    nbytes = np.concatenate([o.line_bytes for o in outputs] + [np.zeros(0, np.int64)])
    order = np.lexsort((np.arange(pos.shape[0]), pos, chrom))           # stable in shard / site order
    dst = np.empty(pos.shape[0], np.int64)
    dst[order] = np.concatenate([[0], np.cumsum(nbytes[order])[:-1]]) if pos.shape[0] else np.zeros(0, np.int64)
    header = header_of(names).encode("ascii")
    total = int(nbytes.sum())
    with open(output_path, "wb") as fh:
        fh.write(header)
        fh.truncate(len(header) + total)
    if total == 0:
        return 0
    out = np.memmap(output_path, dtype=np.uint8, mode="r+", offset=len(header), shape=(total,))
    at = 0
    for o in outputs:
        n = int(o.position.shape[0])
        if n == 0:
            continue
        blob = np.fromfile(o.prefix + ".mean.vcf", dtype=np.uint8)
        lens = o.line_bytes
        if int(lens.sum()) != blob.shape[0]:
            raise ValueError(f"{o.prefix}.mean.vcf holds {blob.shape[0]} bytes, its keys describe {int(lens.sum())}")
        src0 = np.concatenate([[0], np.cumsum(lens)[:-1]])
        out[np.repeat(dst[at:at + n] - src0, lens) + np.arange(blob.shape[0])] = blob
        at += n
    out.flush()
    del out
    return int(pos.shape[0])
