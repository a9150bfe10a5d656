"""One site per call -- the reference's deployment form (caller_calling.py:872-891) -- on one engine: host time per call
and, under rocprofv3, the kernel chain of a call.

    python tools/one_site_profile.py [--calls 400] [--coverage 30] [--per-op]
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_1site -- python3 tools/one_site_profile.py

Every call is Engine.forward(reads, reads_per_allele, alleles_per_site, posteriors=True) with NumPy arrays in and NumPy
logits + posteriors out (synchronous): H2D copy, featurised reads -> read convolver -> allele stage -> posteriors, D2H.
--per-op adds the engine's own HIP-event clock per op (a separate pass: events serialise the launches).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hello_amd import netspec as ns, synth, weights  # noqa: E402
from hello_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=400)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--config", default="single_tech")
    ap.add_argument("--per-op", action="store_true")
    args = ap.parse_args()
    spec = ns.build(args.config)
    eng = Engine(spec, weights.synth_state(spec, seed=1), device=0)
    hybrid = dict(hybrid_coverage=15) if spec.has("read_convolver1") or spec.has("readConv1") else {}
    batch = synth.make_sites(64, seed=3, coverage=args.coverage, channels=eng.program.channels0, **hybrid)
    sites = [batch.site_slice(s, s + 1) for s in range(batch.n_sites)]
    for s in sites[:8]:
        eng.forward_batch(s, posteriors=True)
    t0 = time.perf_counter()
    for i in range(args.calls):
        eng.forward_batch(sites[i % len(sites)], posteriors=True)
    dt = time.perf_counter() - t0
    print(f"{args.config}: {args.calls} one-site calls, {1e3 * dt / args.calls:.4f} ms per call "
          f"({np.mean([s.reads0.shape[0] for s in sites]):.1f} reads, {np.mean([len(s.reads_per_allele0) for s in sites]):.2f} alleles per site)")
    if args.per_op:
        eng.set_profiling(64)
        for i in range(64):
            eng.forward_batch(sites[i % len(sites)], posteriors=True)
        rows, n = eng.op_times_ms()
        for kind, name, ms in rows:
            print(f"    {1e3 * ms:8.1f} us  {kind:18s} {name}")
        print(f"    {1e3 * sum(r[2] for r in rows):8.1f} us  sum of the ops' device times over {n} calls")
    eng.close()


if __name__ == "__main__":
    main()
