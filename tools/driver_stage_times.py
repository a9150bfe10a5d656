"""Where the time of the scoring-stage driver (hello_amd.call.score_shard) goes on a synthetic shard: packing the
aligned reads, the featurizer launch, the forward, and the per-site genotype / VCF / .features records.

    python tools/driver_stage_times.py [--sites 4000] [--coverage 30]

The engine's batched rate (bench.py) is set by the GPU; this shows what the Python host side around it costs per site,
i.e. how many host processes the driver needs in front of one GPU (the reference runs one per core, call.py:26-30,111).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hello_amd import call as driver, featurizer, netspec as ns, shards, vcf, weights  # noqa: E402
from hello_amd.featurizer import AlignedRead  # noqa: E402
from hello_amd.wrapper import ScoringNetwork, pair_keys  # noqa: E402


BAM_CMATCH = 0                                   # pysam / BAM CIGAR operation code of an alignment match


def synth_sites(rng, n, coverage):
    sites = []
    for s in range(n):
        window_start = 1000 + 700 * s
        ref_len = 520
        reference = "".join(rng.choice(list("ACGT"), size=ref_len))
        start = window_start + 240 + int(rng.integers(0, 20))
        ref_allele = reference[start - window_start]
        alt = "ACGT"[("ACGT".index(ref_allele) + 1 + int(rng.integers(0, 3))) % 4]
        per_allele = np.maximum(1, rng.multinomial(max(2, rng.poisson(coverage)), [0.5, 0.5]))
        alleles = []
        for a, count in zip((ref_allele, alt), per_allele):
            reads = []
            for _ in range(int(count)):
                n_bases = 150
                st = start - int(rng.integers(20, 130))
                reads.append(AlignedRead("".join(rng.choice(list("ACGT"), size=n_bases)), rng.integers(2, 60, size=n_bases).tolist(),
                                         [(BAM_CMATCH, n_bases)], st, mapq=int(rng.integers(0, 80)),
                                         orientation=int(rng.choice([-1, 1])), hp=0))
            alleles.append((a, reads, None))
        sites.append(shards.CandidateSite("chr1", start, start + 1, reference, window_start, alleles))
    return sites


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=4000)
    ap.add_argument("--coverage", type=int, default=30)
    args = ap.parse_args()
    import torch
    rng = np.random.default_rng(7)
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=1)
    net = ScoringNetwork(spec, state, device=0, providePredictions=True)
    eng = net.engine
    t = time.perf_counter()
    sites = synth_sites(rng, args.sites, args.coverage)
    t_synth = time.perf_counter() - t
    n_reads = sum(len(r0) for s in sites for _, r0, _ in s.alleles)
    driver.score_shard(net, sites[:64])                                   # warm-up (allocations, first launches)
    packed_shard = shards.PackedShard.from_sites(sites)                    # what PackedShard.from_file yields for a shard file

    t0 = time.perf_counter()
    site_reads = [s.site_reads(0) for s in sites]
    t1 = time.perf_counter()
    packed = featurizer.pack_sites(site_reads)
    t2 = time.perf_counter()
    dev0, rpa0, aps = featurizer.featurize(eng, site_reads, 150, False, device_output=True)     # packs again inside
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    logits, meta, post = eng.forward(dev0, rpa0, aps, None, None, None, posteriors=True)
    post = post.cpu().numpy()
    t4 = time.perf_counter()
    out, col = [], 0
    for site in sites:
        keys = pair_keys([a for a, _, _ in site.alleles])
        n = len(keys)
        rows = [dict(zip(keys, (float(v) for v in post[r, col:col + n]))) for r in range(4)]
        col += n
        ref = driver.WindowReference(site.reference, site.window_start)
        c = vcf.call_site(rows[0], site.chromosome, site.start, site.stop - site.start, ref, info="MixtureOfExpertPrediction")
        if c is not None:
            out.append((c.line(), vcf.feature_record((rows[0], rows[1], rows[2], rows[3], np.array([1.0, 0.0, 0.0], np.float32)),
                                                     site.chromosome, site.start, site.stop - site.start)))
    t5 = time.perf_counter()
    del packed
    t6 = time.perf_counter()
    driver.score_shard(net, sites)
    t7 = time.perf_counter()
    arrays = packed_shard.featurizer_arrays(0)
    t8 = time.perf_counter()
    driver.score_shard(net, packed_shard)
    t9 = time.perf_counter()
    del arrays
    repeats = []
    for _ in range(4):
        ta = time.perf_counter()
        driver.score_shard(net, packed_shard)
        repeats.append(time.perf_counter() - ta)
    us = lambda dt: 1e6 * dt / args.sites                                 # noqa: E731
    print(f"{args.sites} sites, {n_reads} reads ({n_reads / args.sites:.1f} per site); synthesis {t_synth:.2f} s (not a stage)")
    print(f"  site_reads() views            {us(t1 - t0):8.1f} us/site")
    print(f"  pack_sites (host arrays)      {us(t2 - t1):8.1f} us/site")
    print(f"  featurize (pack + launch)     {us(t3 - t2):8.1f} us/site")
    print(f"  forward + posteriors to host  {us(t4 - t3):8.1f} us/site")
    print(f"  genotype / VCF / .features    {us(t5 - t4):8.1f} us/site   ({len(out)} records)")
    print(f"  score_shard from site objects {us(t7 - t6):8.1f} us/site = {args.sites / (t7 - t6):,.0f} sites/s per host process")
    print(f"  PackedShard.featurizer_arrays {us(t8 - t7):8.1f} us/site")
    print(f"  score_shard from a PackedShard{us(t9 - t8):8.1f} us/site = {args.sites / (t9 - t8):,.0f} sites/s per host process")
    print("  ... four more times            " + ", ".join(f"{us(r):.1f}" for r in repeats) +
          f" us/site (best {args.sites / min(repeats):,.0f} sites/s)")
    net.close()


if __name__ == "__main__":
    main()
