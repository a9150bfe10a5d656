"""Soak of the laned program (small launches of multi-chain models on concurrent streams): N random launches of 1..12 sites through an
engine that runs lanes and through one that runs the sequential program; logits, meta weights and pair posteriors must be equal bit
for bit on every launch (a missing event between lanes would show as a changed bit sooner or later).

    python tools/lanes_soak.py [--config hybrid_full] [--launches 3000]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from hello_amd import compiler, netspec as ns, synth, weights
    from hello_amd.engine import Engine
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="hybrid_full")
    ap.add_argument("--launches", type=int, default=3000)
    args = ap.parse_args()
    spec = ns.build(args.config)
    state = weights.synth_state(spec, seed=9)
    laned = Engine(spec, state, device=0)
    sequential = Engine(spec, state, device=0, program=compiler.compile_model(spec, state))
    pool = synth.make_sites(400, seed=11, coverage=(5, 60), hybrid_coverage=(3, 30))
    rng = np.random.default_rng(2)
    bad, t_l, t_s = 0, 0.0, 0.0
    for i in range(args.launches):
        lo = int(rng.integers(0, 388))
        sub = pool.site_slice(lo, lo + int(rng.integers(1, 13)))
        t0 = time.perf_counter()
        a = laned.forward_batch(sub, posteriors=True)
        t1 = time.perf_counter()
        b = sequential.forward_batch(sub, posteriors=True)
        t2 = time.perf_counter()
        t_l, t_s = t_l + (t1 - t0), t_s + (t2 - t1)
        bad += int(not all((x is None) == (y is None) and (x is None or np.array_equal(x, y)) for x, y in zip(a, b)))
    print(f"{args.config}: {args.launches} random launches of 1..12 sites, {laned.lanes_program.n_lanes} lanes: {bad} launches differ from the sequential "
          f"program; mean {1e3 * t_l / args.launches:.3f} ms per launch on lanes, {1e3 * t_s / args.launches:.3f} ms sequential")
    laned.close()
    sequential.close()
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
