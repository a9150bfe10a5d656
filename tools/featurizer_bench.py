#!/usr/bin/env python3
"""Throughput of the GPU pileup-tensor producer on synthetic reads (host inputs; kernel time via rocprofv3)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hello_amd import netspec as ns, weights
from hello_amd.engine import Engine
from hello_amd.featurizer import AlignedRead, SiteReads, featurize

rng = np.random.default_rng(0)
n_sites, reads_per_site = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 30
sites = []
for s in range(n_sites):
    reference = "".join(rng.choice(list("ACGT"), size=600))
    alleles = []
    for k in range(2):
        reads = []
        for _ in range(reads_per_site // 2):
            start = int(rng.integers(200, 330))
            d = int(rng.integers(40, 100))
            cigar = [(0, d), (2, 2), (0, 150 - d - 3), (1, 3)] if rng.random() < 0.2 else [(0, 150)]
            n = sum(l for o, l in cigar if o in (0, 1, 4))
            reads.append(AlignedRead("".join(rng.choice(list("ACGT"), size=n)), rng.integers(10, 41, size=n).tolist(),
                                     cigar, start, mapq=int(rng.integers(5, 61)), orientation=int(rng.choice([-1, 1]))))
        alleles.append((f"a{k}", reads))
    sites.append(SiteReads(reference, 0, 300, 304, alleles))
spec = ns.build("single_tech")
eng = Engine(spec, weights.synth_state(spec, seed=1))
for _ in range(3):
    t0 = time.perf_counter()
    out, rpa, aps = featurize(eng, sites, device_output=True)
    dt = time.perf_counter() - t0
print(f"{out.shape[0]} reads, {out.numel() / 1e6:.1f} MB out, host-inclusive {dt * 1e3:.1f} ms")
