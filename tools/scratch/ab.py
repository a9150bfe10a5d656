import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from hello_amd import compiler, netspec as ns, synth, weights
from hello_amd.engine import Engine
cfg, mode = sys.argv[1], sys.argv[2]
spec = ns.build(cfg); state = weights.synth_state(spec, seed=1)
eng = Engine(spec, state, device=0) if mode == "lanes" else Engine(spec, state, device=0, program=compiler.compile_model(spec, state))
pool = synth.make_sites(256, seed=3, coverage=30, hybrid_coverage=15)
out = []
for n in (1, 2, 4, 8, 16, 32, 64):
    subs = [pool.site_slice(s, s + n) for s in range(0, 256 - n + 1, n)][:16]
    for i in range(10): eng.forward_batch(subs[i % len(subs)], posteriors=True)
    t = time.perf_counter()
    for i in range(150): eng.forward_batch(subs[i % len(subs)], posteriors=True)
    out.append(f"{n}: {(time.perf_counter()-t)/150*1e3:.3f}")
print(cfg, mode, " ".join(out))
