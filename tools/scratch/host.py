import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hello_amd import compiler, netspec as ns, synth, weights
from hello_amd.engine import Engine
for cfg in ("single_tech", "hybrid_no_ensemble", "hybrid_full"):
    spec = ns.build(cfg); state = weights.synth_state(spec, seed=1)
    for mode in ("lanes", "seq"):
        eng = Engine(spec, state, device=0) if mode == "lanes" else Engine(spec, state, device=0, program=compiler.compile_model(spec, state))
        kw = dict(hybrid_coverage=15) if cfg != "single_tech" else {}
        b = synth.make_sites(1, seed=3, coverage=30, **kw)
        t = lambda x: None if x is None else torch.from_numpy(x).cuda()
        dev = synth.SiteBatch(t(b.reads0), b.reads_per_allele0, b.alleles_per_site, t(b.ref_onehot), t(b.reads1), b.reads_per_allele1)
        out = None
        for i in range(20): eng.forward_batch(dev, posteriors=True)
        torch.cuda.synchronize()
        t_submit = t_total = 0.0
        for i in range(300):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.forward_batch(dev, posteriors=True)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            t_submit += (t1 - t0) / 300
            t_total += (t2 - t0) / 300
        print(f"{cfg:20s} {mode:5s} ops {len(eng.program.ops):3d}: host returns after {t_submit*1e6:7.1f} us, results complete after {t_total*1e6:7.1f} us (one forward at a time)")
        eng.close()
