import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from hello_amd import netspec as ns, synth, weights
from hello_amd.engine import Engine
spec = ns.build("single_tech"); state = weights.synth_state(spec, seed=1)
b = synth.make_sites(8192, seed=1001, coverage=30)
r = torch.from_numpy(b.reads0).cuda()
for arith in ("fp32", "bf16x3"):
    eng = Engine(spec, state, device=0, arithmetic=arith)
    for _ in range(3): eng.forward(r, b.reads_per_allele0, b.alleles_per_site, posteriors=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): out = eng.forward(r, b.reads_per_allele0, b.alleles_per_site, posteriors=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    eng.set_profiling(10)
    for _ in range(10): eng.forward(r, b.reads_per_allele0, b.alleles_per_site, posteriors=True)
    torch.cuda.synchronize(); rows, n = eng.op_times_ms(); eng.set_profiling(0)
    print(arith, f"{dt*1e3:.3f} ms/launch = {8192/dt:,.0f} sites/s; readconv {rows[0][2]:.3f} ms; logits[:4]", out[0][0,:4].cpu().numpy())
    eng.close()
