"""Randomised cross-check on the GPU: the fused Winograd engine against the layer-by-layer direct-form engine
(two independent kernel paths) on many random batch shapes, plus run-to-run bit reproducibility.

    python tools/stress_parity.py [--rounds 120] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from hello_amd import netspec as ns, synth, weights
    from hello_amd.engine import Engine
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    cases = [("single_tech", dict()), ("single_tech_hp", dict(channels=7, tech="pacbio")),
             ("hybrid_no_ensemble", dict(hybrid_coverage=12)), ("hybrid_full", dict(hybrid_coverage=9)),
             ("single_tech_addendum", dict()), ("merged_hybrid", dict(hybrid_coverage=10)),
             ("merged_hybrid_250", dict(hybrid_coverage=8, window=250)), ("single_tech_softplus", dict())]
    engines = {}
    worst, t0 = 0.0, time.time()
    for r in range(args.rounds):
        cfg, kw = cases[r % len(cases)]
        if cfg not in engines:
            spec = ns.build(cfg)
            state = weights.synth_state(spec, seed=77)
            engines[cfg] = (Engine(spec, state, device=0, fused=True, winograd=True),
                            Engine(spec, state, device=0, fused=False, winograd=False))
        fast, slow = engines[cfg]
        n_sites = int(rng.choice([1, 2, 3, 5, 17, 64, 255, 256, 257, 700, 2048, 3000]))
        options = [1, 2, 5, 30, (8, 52), (20, 80), 200] if n_sites <= 700 else [5, 30, (8, 52)]
        cov = options[int(rng.integers(len(options)))]
        batch = synth.make_sites(n_sites, seed=int(rng.integers(1 << 30)), coverage=cov, **kw)
        a, am, ap_ = fast.forward_batch(batch, posteriors=True)
        a2, _, ap2 = fast.forward_batch(batch, posteriors=True)
        b, bm, bp = slow.forward_batch(batch, posteriors=True)
        assert np.array_equal(a, a2) and np.array_equal(ap_, ap2), f"round {r}: not reproducible"
        assert np.isfinite(a).all() and np.isfinite(ap_).all()
        d = float(np.abs(ap_ - bp).max())
        dl = float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
        worst = max(worst, d)
        assert d < 1e-4 and dl < 1e-4, f"round {r} {cfg} sites={n_sites} cov={cov}: posterior delta {d}, logits {dl}"
        if r % 20 == 0:
            print(f"round {r:4d} {cfg:22s} sites={n_sites:5d} reads={batch.reads0.shape[0]:7d} max|dpost|={d:.2e}", flush=True)
    print(f"{args.rounds} rounds OK, worst posterior delta {worst:.2e}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
