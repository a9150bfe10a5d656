"""Throughput driver with the reference caller's command line.

    python -m hello_amd.call --network model.wrapper.dnn --workdir out --shards 'shards/*.npz' \\
                             [--ref genome.fa] [--num_threads 8] [--include_hp] [--ibam ... --pbam ...]

The reference's ``call.py`` (python/call.py:88-243) detects hotspots, shards them, runs ``caller_calling.main`` per
shard in a process pool -- one site at a time: featurise (C++), score (torch CPU), emit a VCF line, collect the
``.features`` entry (caller_calling.py:859-900) -- checks every shard log for the sentinel ``Completed running the
script`` (:225-229) and lets ``prepareVcf.main`` build the final VCF from the ``.features`` files (:233-241).

This driver keeps that contract from the point where a shard's candidate sites exist (``hello_amd.shards``: BAM /
FASTA ingestion, hotspot detection and allele assembly need pysam and the C++ searcher and stay upstream), and runs
everything after it on the GPU, a whole shard per launch:

    reads + CIGARs --hello_engine_featurize--> uint8 pileups (device) --Engine.forward--> logits, meta, pair
    posteriors --> per site: record line (caller_calling.py:698-743) + ``.features`` entry (:743-754)

and writes, per shard N, ``<workdir>/<features dir>/features<N>.vcf``, ``features<N>.features`` (the pickle
``prepareVcf`` reads) and ``features<N>.log`` ending in the sentinel; then ``<workdir>/results.output.vcf`` from the
meta-weighted mean of the experts (prepareVcf.py:126-176,199-260; sorted in process instead of by ``vcf-sort``).
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
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import shards as shard_io
from . import vcf
from .wrapper import pair_keys

SENTINEL = "Completed running the script"            # caller_calling.py:902, checked by call.py:225-229
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
    seg = ref[left:left + span]
    out = np.zeros((span, 5), np.uint8)
    out[np.arange(len(seg)), ["ACGT".find(b) if b in "ACGT" else 4 for b in seg]] = 1
    return out


# ------------------------------------------------------------------------------------------------
# one shard
# ------------------------------------------------------------------------------------------------
def score_shard(network, sites, include_hp: bool = False, genomes: Optional[Dict[str, str]] = None,
                feature_length: int = FEATURE_LENGTH, keep=None):
    """Featurise and score every site of a shard in one launch each.  -> [(record line | None, .features entry |
    None)] per site, in order -- what caller_calling.vcfRecords returns per site (:657-754).  ``sites``: a
    ``shards.PackedShard`` (the flat arrays of a shard file: no Python object per read) or a list of
    ``shards.CandidateSite``; ``keep``: optional per-site booleans, sites marked False are scored but emit nothing."""
    import torch
    from .featurizer import featurize
    eng = network.engine
    prog = eng.program
    hybrid = bool(prog.channels1)
    packed = sites if isinstance(sites, shard_io.PackedShard) else shard_io.PackedShard.from_sites(sites)
    if packed.n_sites == 0:
        return []
    if hybrid and not packed.has_reads(1):
        raise ValueError("this model scores two read technologies: every allele of the shard needs both read sets")
    want0 = 7 if include_hp else 6
    if prog.channels0 != want0:
        raise ValueError(f"--include_hp {'set' if include_hp else 'not set'}: the featurizer would write {want0} channels, "
                         f"the model reads {prog.channels0}")
    dev0, rpa0, aps = featurize(eng, packed.featurizer_arrays(0), feature_length, include_hp, device_output=True)
    dev1 = rpa1 = None
    if hybrid:
        dev1, rpa1, _ = featurize(eng, packed.featurizer_arrays(1), feature_length, prog.channels1 == 7, device_output=True)
    n = packed.n_sites
    starts, stops = packed.start.tolist(), packed.stop.tolist()
    refs = [genomes[c] if genomes and c in genomes else WindowReference(packed.reference(s), int(packed.window_start[s]))
            for s, c in enumerate(packed.chromosomes)]
    seg = None
    if prog.uses_ref:
        seg = torch.from_numpy(np.stack([reference_segment(refs[s], starts[s], stops[s], feature_length)
                                         for s in range(n)])).to(dev0.device)
    logits, meta, post = eng.forward(dev0, rpa0, aps, dev1, rpa1, seg, posteriors=True)
    post = post.cpu().numpy().astype(np.float64).tolist()      # float(np.float32) == the same double
    meta = meta.cpu().numpy() if meta is not None else None
    no_meta = np.array([1.0, 0.0, 0.0], np.float32)
    out, col = [], 0
    for s in range(n):
        keys = pair_keys(packed.names(s))
        k = len(keys)
        rows = [dict(zip(keys, post[r][col:col + k])) for r in range(4)]
        col += k
        if keep is not None and not keep[s]:
            out.append((None, None))
            continue
        length = stops[s] - starts[s]
        chromosome = packed.chromosomes[s]
        call = vcf.call_site(rows[0], chromosome, starts[s], length, refs[s], info="MixtureOfExpertPrediction")
        if call is None:                               # no alternative allele at the site: nothing is written (:720-721)
            out.append((None, None))
            continue
        m = meta[s] if meta is not None else no_meta
        out.append((call.line(), vcf.feature_record((rows[0], rows[1], rows[2], rows[3], m), chromosome, starts[s], length)))
    return out


def run_shard(network, shard_path: str, output_prefix: str, include_hp: bool, genomes, loaded=None, keep=None) -> Tuple[str, str]:
    """caller_calling.main for one shard (:757-904): -> (features file, log file)."""
    log_path = output_prefix + ".log"
    t0 = time.perf_counter()
    with open(log_path, "w") as log:
        sites = loaded if loaded is not None else shard_io.PackedShard.from_file(shard_path)
        log.write(f"Shard {shard_path}: {len(sites)} candidate sites\n")
        results = score_shard(network, sites, include_hp, genomes, keep=keep)
        features = []
        with open(output_prefix + ".vcf", "w") as fh:
            for i, (line, feats) in enumerate(results):
                if line is not None:
                    fh.write(line + "\n")
                    features.append(feats)
                if (i + 1) % 100 == 0:
                    log.write("Completed %d sites\n" % (i + 1))
        vcf.write_features(output_prefix + ".features", features)
        log.write(f"Scored {len(sites)} sites, {len(features)} records in {time.perf_counter() - t0:.3f} s\n")
        log.write(SENTINEL + "\n")
    return output_prefix + ".features", log_path


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
    ap.add_argument("--num_threads", type=int, default=30, help="host threads reading shards ahead of the GPU")
    ap.add_argument("--reconcilement_size", default=10, type=int, help="(upstream stage; accepted)")
    ap.add_argument("--include_hp", default=False, action="store_true", help="Include HP tags in tensors")
    ap.add_argument("--shards", required=False,
                    help="glob (or directory) of pre-extracted candidate-site shards (hello_amd.shards), one per "
                         "reference shard<N>.txt")
    ap.add_argument("--device", type=int, default=0)
    return ap


def main(args) -> str:
    logger = logging.getLogger("hello_amd.call")
    if not args.shards:
        raise SystemExit("--shards is required: BAM / FASTA ingestion, hotspot detection and allele assembly are upstream of "
                         "this engine (SURVEY.md section 2, rows 9-13); extract candidate-site shards with the reference's "
                         "searcher and hello_amd.shards.write_shard")
    shard_paths = sorted(glob.glob(os.path.join(args.shards, "*.npz")) if os.path.isdir(args.shards) else glob.glob(args.shards))
    if not shard_paths:
        raise SystemExit(f"no shard matches {args.shards!r}")
    features_dir = os.path.join(args.workdir, features_dir_name(args.ibam, args.pbam))
    os.makedirs(features_dir, exist_ok=True)
    wanted = set(args.chromosomes.split(",")) if args.chromosomes else None
    genomes = read_fasta(args.ref, wanted) if args.ref else {}

    from .loader import load
    network = load(args.network, device=args.device)
    network.eval()
    network.providePredictions = True                  # caller_calling.py:865-868

    feature_files, logs, windows = [], [], {}
    # host threads read shards ahead of the GPU, which scores them one launch at a time, in order
    with ThreadPoolExecutor(max_workers=max(1, min(args.num_threads, 8))) as pool:
        loaded = [pool.submit(shard_io.PackedShard.from_file, p) for p in shard_paths]
        for n, (path, fut) in enumerate(zip(shard_paths, loaded)):
            sites = fut.result()
            keep = None if wanted is None else [c in wanted for c in sites.chromosomes]     # --chromosomes
            for s in range(sites.n_sites):
                if keep is None or keep[s]:
                    windows[(sites.chromosomes[s], int(sites.start[s]))] = (sites.reference(s), int(sites.window_start[s]))
            ff, lg = run_shard(network, path, os.path.join(features_dir, "features%d" % n), args.include_hp, genomes, sites, keep)
            feature_files.append(ff)
            logs.append(lg)
            logger.info("Completed shard %d of %d (%d sites)", n + 1, len(shard_paths), len(sites))
    network.close()
    for lg in logs:                                     # call.py:225-229
        if SENTINEL not in open(lg).read():
            raise ValueError("Did not run: log file %s doesn't have termination string" % lg)

    def genome_of(rec):
        if rec["chromosome"] in genomes:
            return genomes[rec["chromosome"]]
        return WindowReference(*windows[(rec["chromosome"], rec["position"])])
    result = prepare_vcf(feature_files, os.path.join(args.workdir, "results.output.vcf"), genome_of,
                         {c: len(g) for c, g in genomes.items()})
    logger.info("Completed runs. Results in %s", result)
    return result


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s:%(message)s")
    main(parser().parse_args())
    sys.exit(0)
