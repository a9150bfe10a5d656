"""PCIe-inclusive throughput: host-resident (pinned) pileups -> host-resident logits + posteriors through
hello_amd.pipeline.HostPipeline, next to the serial host path of Engine.forward.  One JSON line.

    python tools/host_pipeline_bench.py [--sites 8192] [--steps 20] [--depth 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from hello_amd import netspec as ns, synth, weights
    from hello_amd.engine import Engine
    from hello_amd.pipeline import HostPipeline
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--pool", type=int, default=3)
    ap.add_argument("--engines", type=int, default=1, help="engines (compute streams) the pipeline alternates over")
    args = ap.parse_args()
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=1)
    eng = Engine(spec, state, device=0)
    engines = [eng] + [Engine(spec, state, device=0) for _ in range(args.engines - 1)]
    pool = []
    for i in range(args.pool):
        b = synth.make_sites(args.sites, seed=2000 + i, coverage=30)
        pinned = synth.SiteBatch(torch.from_numpy(b.reads0).pin_memory(), b.reads_per_allele0, b.alleles_per_site,
                                 b.ref_onehot)
        pool.append((b, pinned))
    mb = pool[0][0].reads0.nbytes / 1e6

    def run(kind):
        pipe = HostPipeline(depth=args.depth, engines=engines)
        done = 0
        t0 = time.perf_counter()
        for i in range(args.steps):
            pageable, pinned = pool[i % len(pool)]
            if kind == "serial":
                eng.forward_batch(pageable, posteriors=True)
                done += 1
            else:
                done += len(pipe.submit(pinned if kind == "pinned" else pageable, tag=i))
        done += len(pipe.flush())
        torch.cuda.synchronize()
        assert done == args.steps
        return args.sites * args.steps / (time.perf_counter() - t0)

    res = {}
    for kind in ("serial", "pageable", "pinned"):
        run(kind)                                  # warm-up: allocations, page-ins
        res[kind] = run(kind)
    print(json.dumps({
        "metric": "candidate sites/sec, host-resident inputs and outputs (PCIe inclusive), 1 x MI355X",
        "sites_per_step": args.sites, "steps": args.steps, "input_MB_per_step": round(mb, 1),
        "engine_forward_on_host_arrays": round(res["serial"], 1),
        "pipeline_pageable_inputs": round(res["pageable"], 1),
        "pipeline_pinned_inputs": round(res["pinned"], 1), "depth": args.depth, "engines": args.engines}))


if __name__ == "__main__":
    main()
