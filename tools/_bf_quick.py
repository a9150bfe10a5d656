import sys, time, os; sys.path.insert(0, '.')
import numpy as np, torch
from hello_amd import netspec as ns, synth, weights
from hello_amd.engine import Engine
spec = ns.build("single_tech"); state = weights.synth_state(spec, seed=1)
b = synth.make_sites(8192, seed=1001, coverage=30)
r = torch.from_numpy(b.reads0).cuda()
for arith in sys.argv[1:] or ["bf16x3"]:
    eng = Engine(spec, state, device=0, arithmetic=arith)
    for _ in range(3): eng.forward(r, b.reads_per_allele0, b.alleles_per_site, posteriors=True)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        eng.set_profiling(10)
        for _ in range(10): eng.forward(r, b.reads_per_allele0, b.alleles_per_site, posteriors=True)
        torch.cuda.synchronize(); rows, n = eng.op_times_ms(); eng.set_profiling(0)
        best = min(best, rows[0][2])
    print(os.environ.get("HELLO_LIB", "default"), arith, f"readconv {best:.3f} ms; sum of ops {sum(x[2] for x in rows):.3f} ms")
    eng.close()
