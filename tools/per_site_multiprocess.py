"""The reference's deployment form on one GPU: N worker processes, each with its own engine, each scoring ONE site per
call through the plug-in surface (python/call.py:26-30,111: a process pool, torch single-threaded per worker).

    python tools/per_site_multiprocess.py [--workers 4] [--calls 2000]

Prints each worker's rate and the aggregate.  At most 6 processes may use the card on the shared pool.
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, calls, start_evt, out_q):
    import numpy as np  # noqa: F401
    import torch
    from hello_amd import netspec as ns, synth, weights
    from hello_amd.wrapper import ScoringNetwork
    torch.set_num_threads(1)
    spec = ns.build("single_tech")
    net = ScoringNetwork(spec, weights.synth_state(spec, seed=1), device=0, providePredictions=True)
    batch = synth.make_sites(256, seed=3 + rank, coverage=30)
    sites, r, a = [], 0, 0
    for s in range(batch.n_sites):
        fd = {}
        for k in range(int(batch.alleles_per_site[s])):
            n = int(batch.reads_per_allele0[a])
            fd["A" * (k + 1)] = (torch.from_numpy(batch.reads0[r:r + n]).float(), None)
            r += n
            a += 1
        sites.append((fd, torch.zeros(1, 150, 5)))
    for fd, seg in sites[:32]:
        net(fd, seg)
    out_q.put(("ready", rank, 0.0))
    start_evt.wait()
    t = time.perf_counter()
    for i in range(calls):
        fd, seg = sites[i % len(sites)]
        net(fd, seg)
    dt = time.perf_counter() - t
    out_q.put(("done", rank, calls / dt))
    net.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--calls", type=int, default=2000)
    args = ap.parse_args()
    if args.workers > 6:
        raise SystemExit("at most 6 processes may use the card on this pool")
    ctx = mp.get_context("spawn")
    start_evt, q = ctx.Event(), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, args.calls, start_evt, q)) for r in range(args.workers)]
    for p in procs:
        p.start()
    for _ in procs:
        assert q.get(timeout=300)[0] == "ready"
    start_evt.set()
    rates = sorted((q.get(timeout=600) for _ in procs), key=lambda x: x[1])
    for p in procs:
        p.join()
    for _, rank, rate in rates:
        print(f"  worker {rank}: {rate:8.0f} sites/s ({1e3 / rate:.3f} ms per call)")
    print(f"{args.workers} worker processes, one site per call each: {sum(r[2] for r in rates):,.0f} sites/s in aggregate")


if __name__ == "__main__":
    main()
