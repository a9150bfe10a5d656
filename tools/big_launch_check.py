"""One launch far larger than the bench's 8 192 sites -- up to ~98 k sites / 2.9 M reads / 2.65 GB of pileups, past 2^31 bytes of input --
must give what the same sites give in 8 192-site launches (logits to re-association of the per-allele partial slots, ~5e-7 of scale;
posteriors to ~4e-7): the kernels' 32-bit tile / buffer-descriptor arithmetic and the engine's 64-bit row arithmetic hold at sizes a
caller coming from the reference's DataLoader might hand over in one call.

    python tools/big_launch_check.py
"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from hello_amd import netspec as ns, synth, weights
from hello_amd.engine import Engine
spec = ns.build("single_tech")
eng = Engine(spec, weights.synth_state(spec, seed=1), device=0)
base = [synth.make_sites(8192, seed=100 + i, coverage=30) for i in range(2)]
def cat(bs):
    return synth.SiteBatch(np.concatenate([b.reads0 for b in bs]), np.concatenate([b.reads_per_allele0 for b in bs]),
                           np.concatenate([b.alleles_per_site for b in bs]), np.concatenate([b.ref_onehot for b in bs]), None, None)
parts = [eng.forward_batch(b, posteriors=True) for b in base]
for n in (2, 4, 8, 12):
    big = cat([base[i % 2] for i in range(n)])
    t = time.perf_counter()
    try:
        lg, _, po = eng.forward_batch(big, posteriors=True)
    except Exception as e:
        print(n * 8192, "sites: refused:", repr(e)[:300]); continue
    dt = time.perf_counter() - t
    want = np.concatenate([parts[i % 2][0] for i in range(n)], axis=1)
    wantp = np.concatenate([parts[i % 2][2] for i in range(n)], axis=1)
    scale = np.abs(want).max()
    print(n * 8192, "sites,", big.reads0.shape[0], "reads,", round(big.reads0.nbytes / 1e9, 2), "GB in:", round(dt, 3), "s; logits max|d|/scale", float(np.abs(lg - want).max() / scale),
          "posteriors max|d|", float(np.abs(po - wantp).max()), "finite", bool(np.isfinite(lg).all()))
eng.close()
