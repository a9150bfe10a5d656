"""Throughput driver with the reference caller's command line.

    python -m hello_amd.call --network model.wrapper.dnn --workdir out --shards 'shards/*.npz' \\
                             [--ref genome.fa] [--num_threads 8] [--include_hp] [--ibam ... --pbam ...]

The reference's ``call.py`` (python/call.py:88-243) detects hotspots, shards them, runs ``caller_calling.main`` per
shard in a process pool -- one site at a time: featurise (C++), score (torch CPU), emit a VCF line, collect the
``.features`` entry (caller_calling.py:859-900) -- checks every shard log for the sentinel ``Completed running the
script`` (:225-229) and lets ``prepareVcf.main`` build the final VCF from the ``.features`` files (:233-241).

This driver keeps that contract from the point where a shard's candidate sites exist (``hello_amd.shards``: BAM /
FASTA ingestion, hotspot detection and allele assembly need pysam and the C++ searcher and stay upstream), and runs
everything after it at the engine's rate (``hello_amd.shard_pipeline``): reader threads load shards with bounded
read-ahead; several shards are coalesced into one GPU launch

    reads + CIGARs --hello_engine_featurize--> uint8 pileups (device) --hello_engine_forward--> logits, meta, pair
    posteriors --hello_site_records (host threads)--> record lines (caller_calling.py:698-743), ``.features`` entries
    (:743-754), final-VCF lines (prepareVcf.py:138-168)

while the previous launch's records are written and the next one's bytes are staged.  Per shard N it writes
``<workdir>/<features dir>/features<N>.vcf``, ``features<N>.features`` (the pickle ``prepareVcf`` reads),
``features<N>.log`` ending in the sentinel and ``features<N>.mean.vcf``; then ``<workdir>/results.output.vcf`` from the
meta-weighted mean of the experts (prepareVcf.py:126-176,199-260; sorted in process instead of by ``vcf-sort``).

``--gpus N`` (or a launch under ``python -m torch.distributed.run``) runs one process per GPU: the shard files are dealt
to the ranks as contiguous runs balanced by read count (``hello_amd.shard.partition_sites`` over per-shard read
totals), every rank writes its shards' files, and after ONE barrier rank 0 merges the final VCF from the files and the
ranks' key indexes -- the ranks exchange no pileup or posterior data (the reference's pool does not either,
call.py:215-221).
Flags of the reference that configure upstream stages (--hybrid_hotspot, --q_threshold, --mapq_threshold,
--reconcilement_size, --chromosomes) are accepted so existing command lines keep working; --ibam / --pbam only name
the features directory the way ``get_workdir`` does (call.py:40-48).
"""
from __future__ import annotations

import argparse
import glob
import logging
import os
import pickle
import re
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import shards as shard_io
from . import vcf
from .wrapper import pair_keys

from .shard_pipeline import SENTINEL                 # caller_calling.py:902, checked by call.py:225-229  # noqa: E402
FEATURE_LENGTH = 150                                 # call.py:187


# ------------------------------------------------------------------------------------------------
# reference sequence: a FASTA read as text (no pysam), or the windows the shards carry
# ------------------------------------------------------------------------------------------------
def read_fasta(path: str, wanted: Optional[Sequence[str]] = None) -> Dict[str, str]:
    """chromosome -> sequence (upper / lower case kept).  ``wanted`` limits what is held in memory."""
    keep = set(wanted) if wanted is not None else None
    out: Dict[str, List[str]] = {}
    name = None
    with open(path) as fh:
        for line in fh:
            if line.startswith(">"):
                name = line[1:].split()[0]
                if keep is not None and name not in keep:
                    name = None
                else:
                    out[name] = []
            elif name is not None:
                out[name].append(line.strip())
    return {k: "".join(v) for k, v in out.items()}


class WindowReference:
    """Genome coordinates served from one site's reference window (what the shard carries when no FASTA is given):
    ``ref[a:b]`` / ``ref[i]`` like a chromosome string."""

    def __init__(self, window: str, window_start: int):
        self.window, self.start = window, window_start

    def __getitem__(self, index):
        if isinstance(index, slice):
            lo, hi = index.start - self.start, index.stop - self.start
            if lo < 0 or hi > len(self.window):
                raise IndexError(f"[{index.start}, {index.stop}) leaves the site's reference window "
                                 f"[{self.start}, {self.start + len(self.window)})")
            return self.window[lo:hi]
        i = index - self.start
        if i < 0 or i >= len(self.window):
            raise IndexError(f"position {index} leaves the site's reference window")
        return self.window[i]


def reference_segment(ref, start: int, stop: int, span: int = FEATURE_LENGTH) -> np.ndarray:
    """caller_calling.py:53-97 (get_reference_segment + one_hot_encode): uint8 [span, 5], ACGT + other."""
    mid = (start + stop) // 2
    left = mid - span // 2
    seg = ref[left:left + span] if left >= 0 else ""
    if len(seg) != span:            # the reference's one_hot_encode yields a short tensor here and the network call fails
        raise ValueError(f"the {span} bp reference segment around [{start}, {stop}) leaves the sequence (it starts at {left})")
    out = np.zeros((span, 5), np.uint8)
    out[np.arange(len(seg)), ["ACGT".find(b) if b in "ACGT" else 4 for b in seg]] = 1
    return out


# ------------------------------------------------------------------------------------------------
# one shard
# ------------------------------------------------------------------------------------------------
def _genome_bytes(genomes: Optional[Dict[str, str]]) -> Optional[Dict[str, bytes]]:
    return {k: (v if isinstance(v, bytes) else v.encode("ascii")) for k, v in genomes.items()} if genomes else None


def score_shard(network, sites, include_hp: bool = False, genomes: Optional[Dict[str, str]] = None,
                feature_length: int = FEATURE_LENGTH, keep=None):
    """Featurise and score every site of ONE shard in one launch, synchronously (the building block under test; the
    driver itself streams shards through ``shard_pipeline.run``).  -> [(record line | None, .features entry | None)] per
    site, in order -- what caller_calling.vcfRecords returns per site (:657-754).  ``sites``: a ``shards.PackedShard``
    or a list of ``shards.CandidateSite``; ``keep``: optional per-site booleans, sites marked False are scored but emit
    nothing."""
    from . import records as rec
    from .shard_pipeline import ShardScorer, prepare, site_table
    packed = sites if isinstance(sites, shard_io.PackedShard) else shard_io.PackedShard.from_sites(sites, feature_length)
    if packed.n_sites == 0:
        return []
    scorer = getattr(network, "_shard_scorer", None)
    if scorer is None or scorer.L != feature_length:
        scorer = network._shard_scorer = ShardScorer(network, include_hp, feature_length)
    elif scorer.channels[0] != (7 if include_hp else 6):
        raise ValueError(f"--include_hp {'set' if include_hp else 'not set'}: the featurizer would write {7 if include_hp else 6} "
                         f"channels, the model reads {scorer.channels[0]}")
    prepare(packed, scorer.hybrid, scorer.uses_ref)
    done = scorer.submit([packed]) + scorer.flush()
    scored = done[-1]
    table = site_table([packed], _genome_bytes(genomes), keep=None if keep is None else np.asarray(keep, np.uint8))
    with rec.site_records(table, scored.posteriors, scored.meta, None, features=True) as r:
        entries = iter(pickle.loads(bytes(r.features)))
        out = []
        for s in range(packed.n_sites):
            lo, hi = int(r.shard_vcf_off[s]), int(r.shard_vcf_off[s + 1])
            out.append((bytes(r.shard_vcf[lo:hi - 1]).decode("ascii"), next(entries)) if hi > lo else (None, None))
    return out


def shard_number(path: str, fallback: int) -> int:
    """The N of the reference's ``shard<N>.txt`` / ``features<N>`` (call.py:162-221): the digits that end the file's stem."""
    m = re.search(r"(\d+)$", os.path.splitext(os.path.basename(path))[0])
    return int(m.group(1)) if m else fallback


def natural_key(path: str):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]


# ------------------------------------------------------------------------------------------------
# final VCF (prepareVcf.main)
# ------------------------------------------------------------------------------------------------
def header(chromosomes: Sequence[str], lengths: Dict[str, int]) -> str:
    """prepareVcf.py:185-196."""
    s = "##fileformat=VCFv4.1\n"
    for c in chromosomes:
        s += "##contig=<ID=%s,length=%d>\n" % (c, lengths[c]) if c in lengths else "##contig=<ID=%s>\n" % c
    s += '##INFO=<ID=HELLO,Number=1,Type=String,Description="Obtained from HELLO variant caller">\n'
    s += '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
    s += '##FILTER=<ID=FAIL,Description="Failed call">\n'
    s += "#" + "\t".join("CHROM  POS     ID      REF     ALT     QUAL    FILTER  INFO    FORMAT  SAMPLE1".split()) + "\n"
    return s


def prepare_vcf(feature_files: Sequence[str], output_path: str, genomes_for, lengths: Dict[str, int]) -> str:
    """prepareVcf.main (:199-260) for the label the pipeline uses ("output" = calls on the meta-weighted mean of
    the experts, :154-168), records sorted by (chromosome, position) in process (the reference pipes through the
    external ``vcf-sort``).  ``genomes_for(chromosome, position)`` -> a sequence addressable in genome coordinates."""
    calls: List[vcf.Call] = []
    chromosomes = []
    for path in feature_files:
        for rec in pickle.load(open(path, "rb")):
            genome = genomes_for(rec)
            mean = vcf.mean_posteriors(rec["expertPredictions"], [float(m) for m in rec["meta"]])
            call = vcf.call_site(mean, rec["chromosome"], rec["position"], rec["length"], genome)
            if call is not None:
                calls.append(call)
            if rec["chromosome"] not in chromosomes:
                chromosomes.append(rec["chromosome"])
    order = {c: i for i, c in enumerate(sorted(chromosomes))}
    calls.sort(key=lambda c: (order[c.chromosome], c.position))
    with open(output_path, "w") as fh:
        fh.write(header(sorted(chromosomes), lengths))
        for c in calls:
            fh.write(c.line() + "\n")
    return output_path


# ------------------------------------------------------------------------------------------------
# command line (python/call.py:245-323)
# ------------------------------------------------------------------------------------------------
def features_dir_name(ibam: Optional[str], pbam: Optional[str]) -> str:
    """call.py:33-48 (get_bam_string / get_workdir)."""
    def bam_string(bam):
        bam = os.path.abspath(bam)
        name = os.path.split(os.path.split(bam)[0])[-1].replace("/", "__") + "___" + os.path.split(bam)[-1].replace("/", "__")
        return name.replace(".", "__")
    name = "features"
    for bam in (ibam, pbam):
        if bam:
            name += "_" + bam_string(bam)
    return name


def parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="HELLO variant scoring on MI355X, with call.py's command line")
    ap.add_argument("--ibam", help="Illumina BAM file (names the features directory; BAM ingestion is upstream)")
    ap.add_argument("--pbam", help="PacBio BAM file (names the features directory; BAM ingestion is upstream)")
    ap.add_argument("--ref", help="Reference FASTA (optional when the shards carry their reference windows)")
    ap.add_argument("--workdir", required=True, help="Working directory")
    ap.add_argument("--chromosomes", help="Chromosomes to use (comma-separated): shards' sites elsewhere are skipped")
    ap.add_argument("--network", required=True, help="Network path (.wrapper.dnn pickle or native .npz)")
    ap.add_argument("--hybrid_hotspot", default=False, action="store_true", help="(upstream stage; accepted)")
    ap.add_argument("--q_threshold", default=10, type=int, help="(upstream stage; accepted)")
    ap.add_argument("--mapq_threshold", default=10, type=int, help="(upstream stage; accepted)")
    ap.add_argument("--num_threads", type=int, default=30,
                    help="host threads per GPU: shard readers and the record stage share them (the reference's pool size)")
    ap.add_argument("--reconcilement_size", default=10, type=int, help="(upstream stage; accepted)")
    ap.add_argument("--include_hp", default=False, action="store_true", help="Include HP tags in tensors")
    ap.add_argument("--shards", required=False,
                    help="glob (or directory) of pre-extracted candidate-site shards (hello_amd.shards), one per "
                         "reference shard<N>.txt")
    ap.add_argument("--device", type=int, default=0, help="GPU of a single-process run")
    ap.add_argument("--gpus", type=int, default=1,
                    help="processes / GPUs of this node: > 1 re-launches this command under torch.distributed.run "
                         "(a launch that is already under it reads RANK / WORLD_SIZE instead)")
    ap.add_argument("--sites_per_launch", type=int, default=8192, help="shards are coalesced into GPU launches of about this many sites")
    ap.add_argument("--arithmetic", choices=["fp32", "bf16x3", "bf16x3+32"], default="fp32",
                    help="fp32: exact fp32 everywhere (default).  bf16x3: the read convolver's 64-channel trunk on the bf16 matrix "
                         "cores as 3-term splits (~1.3x faster; posteriors move by ~1e-6)")
    return ap


def _argv_of(args) -> List[str]:
    argv: List[str] = []
    for action in parser()._actions:
        if not action.option_strings or action.dest in ("help", "gpus"):
            continue
        value = getattr(args, action.dest, None)
        if value is None or value is False:
            continue
        argv += [action.option_strings[0]] + ([] if value is True else [str(value)])
    return argv


def _launch_ranks(args) -> int:
    """``--gpus N`` from a plain command line: one child process per GPU under torch.distributed.run, started BEFORE this
    process touches the GPU; this process only waits for them."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "hello_amd.call"] + _argv_of(args) + ["--gpus", str(args.gpus)]
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode


def _resident_mb() -> float:
    try:
        with open("/proc/self/statm") as fh:
            return int(fh.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6
    except (OSError, ValueError, IndexError):
        return float("nan")


def shard_read_totals(paths: Sequence[str], threads: int = 4) -> np.ndarray:
    """Reads the featurizer will write per shard (both technologies; dummy reads included): the weight shards are dealt
    to ranks by.  Only the count arrays of each file are read."""
    from concurrent.futures import ThreadPoolExecutor

    def total(path):
        # the second technology's counts weigh in only when the shard says it has one (has_second; a file that carries
        # reads_per_allele1 with has_second = 0 is refused later by PackedShard.validate -- it must not skew the deal first)
        if path.endswith(".npz"):
            with np.load(path, allow_pickle=False) as z:
                got = {k: z[k] for k in ("reads_per_allele0", "reads_per_allele1", "has_second") if k in z.files}
        else:
            got = shard_io.read_flat_arrays(path, ("reads_per_allele0", "reads_per_allele1", "has_second"))
        second = "has_second" in got and int(np.asarray(got["has_second"]).reshape(-1)[0]) != 0
        return sum(int(np.maximum(got[k], 1).sum()) for k in (("reads_per_allele0", "reads_per_allele1") if second else ("reads_per_allele0",))
                   if k in got)
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        return np.array(list(pool.map(total, paths)), np.int64)


def main(args) -> str:
    logger = logging.getLogger("hello_amd.call")
    if not args.shards:
        raise SystemExit("--shards is required: BAM / FASTA ingestion, hotspot detection and allele assembly are upstream of "
                         "this engine (SURVEY.md section 2, rows 9-13); extract candidate-site shards with the reference's "
                         "searcher and hello_amd.shards.write_shard")
    result_path = os.path.join(args.workdir, "results.output.vcf")
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if getattr(args, "gpus", 1) > 1 and not under_launcher:
        rc = _launch_ranks(args)
        if rc != 0:
            raise SystemExit(rc)
        return result_path
    rank, world = (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])) if under_launcher else (0, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", rank))

    shard_paths = sorted((glob.glob(os.path.join(args.shards, "*.hshard")) + glob.glob(os.path.join(args.shards, "*.npz")))
                         if os.path.isdir(args.shards) else glob.glob(args.shards), key=natural_key)
    if not shard_paths:
        raise SystemExit(f"no shard matches {args.shards!r}")
    numbers = [shard_number(p, i) for i, p in enumerate(shard_paths)]
    if len(set(numbers)) != len(numbers):                # names that do not end in distinct numbers: number them in order
        numbers = list(range(len(shard_paths)))
    features_dir = os.path.join(args.workdir, features_dir_name(args.ibam, args.pbam))
    os.makedirs(features_dir, exist_ok=True)
    wanted = set(args.chromosomes.split(",")) if args.chromosomes else None
    genomes = read_fasta(args.ref, wanted) if args.ref else {}

    import torch
    from . import shard as sharding, shard_pipeline as sp
    lo, hi = 0, len(shard_paths)
    device = args.device
    threads = max(2, min(args.num_threads, len(os.sched_getaffinity(0))))
    if world > 1:
        import torch.distributed as dist
        device = local_rank % max(torch.cuda.device_count(), 1)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        threads = max(2, min(args.num_threads, len(sharding.pin_rank(local_rank, local_world, device))))
        # control plane only (shard totals, the end-of-run barrier): no tensor leaves a rank, so gloo serves every backend
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mine = shard_read_totals(shard_paths[rank::world], threads)
        gathered: List = [None] * world
        dist.all_gather_object(gathered, mine)
        totals = np.zeros(len(shard_paths), np.int64)
        for r, part in enumerate(gathered):
            totals[r::world] = part
        lo, hi = sharding.partition_sites(totals, world)[rank]
        logger.info("rank %d of %d: shards [%d, %d) of %d, %d reads of %d, GPU %d, %d host threads", rank, world, lo, hi,
                    len(shard_paths), int(totals[lo:hi].sum()), int(totals.sum()), device, threads)

    from .loader import load
    network = load(args.network, device=device, arithmetic=getattr(args, "arithmetic", "fp32"))
    network.eval()
    network.providePredictions = True                  # caller_calling.py:865-868
    readers = max(1, min(threads // 2, 8))
    t0 = time.perf_counter()
    stats = sp.run(network, shard_paths[lo:hi], lambda n: os.path.join(features_dir, "features%d" % n), args.include_hp,
                   _genome_bytes(genomes), wanted, reader_threads=readers, record_threads=max(1, threads - readers),
                   sites_per_launch=getattr(args, "sites_per_launch", 8192), tags=numbers[lo:hi])
    network.close()
    logger.info("rank %d: %d shards, %d sites, %d reads in %d launches, %.2f s (%.0f sites/s; waiting for readers %.2f s, "
                "staging %.2f s, record stage %.2f s on its threads; resident set after the loop %.0f MB)", rank, hi - lo,
                stats.sites, stats.reads, stats.launches, stats.seconds, stats.sites / max(stats.seconds, 1e-9), stats.wait_read,
                stats.stage_seconds, stats.record_seconds, _resident_mb())
    for out in stats.outputs:                           # call.py:225-229
        if SENTINEL not in open(out.prefix + ".log").read():
            raise ValueError("Did not run: log file %s doesn't have termination string" % (out.prefix + ".log"))
    outputs = stats.outputs
    if world > 1:
        import torch.distributed as dist
        sp.save_index(os.path.join(features_dir, "mean_index.rank%d.npz" % rank), outputs)
        dist.barrier()                                  # THE synchronisation of the run: every rank's files are complete
        if rank == 0:
            outputs = [o for r in range(world) for o in sp.load_index(os.path.join(features_dir, "mean_index.rank%d.npz" % r))]
    if rank == 0:
        lengths = {c: len(g) for c, g in genomes.items()}
        n = sp.merge_final_vcf(outputs, lambda names: header(names, lengths), result_path)
        logger.info("Completed runs in %.2f s. %d records in %s", time.perf_counter() - t0, n, result_path)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return result_path


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s:%(message)s")
    main(parser().parse_args())
    sys.exit(0)
