import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from hello_amd import netspec as ns, synth, weights
from hello_amd.engine import Engine
n = int(sys.argv[1])
spec = ns.build("single_tech")
eng = Engine(spec, weights.synth_state(spec, seed=1), device=0)
batch = synth.make_sites(64, seed=3, coverage=30)
subs = [batch.site_slice(s, s + n) for s in range(0, 64 - n + 1, n)]
for i in range(20):
    s = subs[i % len(subs)]
    eng.forward(s.reads0, s.reads_per_allele0, s.alleles_per_site, posteriors=True)
t = time.perf_counter()
for i in range(200):
    s = subs[i % len(subs)]
    eng.forward(s.reads0, s.reads_per_allele0, s.alleles_per_site, posteriors=True)
print(n, "sites per call:", (time.perf_counter() - t) / 200 * 1e3, "ms")
eng.close()
