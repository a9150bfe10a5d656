"""Per-op device times of one configuration at a full launch, inputs resident in HBM (the engine's own HIP events around every
op, a separate pass from the wall-clock loop), and the forward's logits -- saved, or compared with a saved file, so that kernel
variants (HELLO_LIB=other.so) are timed AND held to the adopted build's answers in one go.

    python tools/allele_stage_bench.py [--config single_tech] [--sites 8192] [--save ref.npy | --check ref.npy]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="single_tech")
    ap.add_argument("--sites", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--arithmetic", default="fp32")
    ap.add_argument("--save")
    ap.add_argument("--check")
    ap.add_argument("--label", default=os.environ.get("HELLO_LIB", "in-tree library"))
    args = ap.parse_args()
    import torch
    from hello_amd import netspec as ns, synth, weights
    from hello_amd.engine import Engine
    spec = ns.build(args.config)
    eng = Engine(spec, weights.synth_state(spec, seed=1), device=0, arithmetic=args.arithmetic)
    kw = dict(coverage=30)
    if spec.hybrid_inputs:
        kw["hybrid_coverage"] = 15
    b = synth.make_sites(args.sites, seed=1001, **kw)
    dev = torch.device("cuda", 0)
    reads0 = torch.from_numpy(b.reads0).to(dev)
    reads1 = torch.from_numpy(b.reads1).to(dev) if b.reads1 is not None else None
    ref = torch.from_numpy(b.ref_onehot).to(dev) if eng.program.uses_ref else None
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        return eng.forward(reads0, b.reads_per_allele0, b.alleles_per_site, reads1, b.reads_per_allele1, ref, stream=stream, posteriors=True)
    for _ in range(3):
        out = step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        out = step()
    torch.cuda.synchronize(dev)
    ms = 1e3 * (time.perf_counter() - t0) / args.reps
    logits = out[0].cpu().numpy()
    eng.set_profiling(10)
    for _ in range(10):
        step()
    torch.cuda.synchronize(dev)
    rows, n = eng.op_times_ms()
    eng.set_profiling(0)
    print(f"== {args.label} [{args.arithmetic}]: {args.config}, {args.sites} sites ({b.reads0.shape[0]} reads, {b.n_alleles} alleles): {ms:.3f} ms per forward "
          f"= {args.sites / ms:.1f} k sites/s device-resident")
    behind = 0.0
    for (kind, name, t), op in zip(rows, eng.program.ops):
        form = "wino" if (op.kind == 1 and op.flags & 32) else ""
        print(f"    {t:8.4f} ms  {kind:16s} {form:5s} {name}")
        behind += 0.0 if op.kind == 8 else t
    print(f"    {behind:8.4f} ms  everything behind the read convolver(s) ({sum(1 for o in eng.program.ops if o.kind != 8)} launches); "
          f"wino layers {sum(t for (k, nm, t), o in zip(rows, eng.program.ops) if o.kind == 1 and o.flags & 32):.4f} ms")
    if args.save:
        np.save(args.save, logits)
    if args.check:
        want = np.load(args.check)
        scale = max(1.0, float(np.abs(want).max()))
        print(f"    logits vs {args.check}: max |d| / scale = {float(np.abs(logits - want).max()) / scale:.3e}, identical bits: {np.array_equal(logits, want)}")
    eng.close()


if __name__ == "__main__":
    main()
