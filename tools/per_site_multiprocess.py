"""The reference's deployment form on one GPU: N worker processes, each scoring ONE site per call through the plug-in surface
(python/call.py:26-30,111: a process pool, torch single-threaded per worker; python/caller_calling.py:863-868,872-891).

    python tools/per_site_multiprocess.py [--workers 4] [--calls 2000]                 # every worker its own engine
    python tools/per_site_multiprocess.py --shared [--workers 16] [--calls 2000]       # loader.load(path, shared=True): ONE server

Prints each worker's rate and the aggregate (and, with --shared, the server's launch statistics).  Without --shared at most 6
processes may use the card on the shared pool; with --shared only the server process does, so the worker count is free.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, calls, start_evt, out_q, model_path, shared, config):
    import numpy as np  # noqa: F401
    import torch
    import bench
    from hello_amd import loader
    torch.set_num_threads(1)
    net = loader.load(model_path, shared=shared, providePredictions=True)
    sites = [({a: (torch.from_numpy(f), None if g is None else torch.from_numpy(g)) for a, (f, g) in fd.items()}, torch.from_numpy(seg))
             for fd, seg in bench.feature_dicts(bench.make_config_sites(config, 256, 3 + rank))]
    for fd, seg in sites[:32]:
        net(fd, seg)
    out_q.put(("ready", rank, 0.0, None))
    start_evt.wait()
    t = time.perf_counter()
    for i in range(calls):
        fd, seg = sites[i % len(sites)]
        net(fd, seg)
    dt = time.perf_counter() - t
    stats = net.server_stats() if shared else None
    if stats is not None:                          # does this worker hold the GPU's device nodes open?  (it must not: only the server does)
        fds = []
        for fd in os.listdir("/proc/self/fd"):
            try:
                fds.append(os.readlink(f"/proc/self/fd/{fd}"))
            except OSError:
                pass
        stats["worker_gpu_fds"] = sorted({f for f in fds if "kfd" in f or "renderD" in f})
    out_q.put(("done", rank, calls / dt, stats))
    net.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--calls", type=int, default=2000)
    ap.add_argument("--shared", action="store_true", help="workers load the model with shared=True: one scoring server for all of them")
    ap.add_argument("--engines", type=int, default=0, help="engines (scorer threads) of the shared server; 0 = the server picks (2; 1 for laned models)")
    ap.add_argument("--config", default="C2", help="bench.py configuration (model + synthetic sites)")
    ap.add_argument("--json", action="store_true", help="print ONE JSON line (bench.py's `per_site_shared` leg) instead of the table")
    args = ap.parse_args()
    if args.workers > 6 and not args.shared:
        raise SystemExit("at most 6 processes may use the card on this pool")
    import bench
    from hello_amd import loader, netspec as ns, weights
    tmp = tempfile.mkdtemp(prefix="hello_per_site_")
    os.environ.setdefault("HELLO_SHARED_DIR", os.path.join(tmp, "rendezvous"))
    os.environ["HELLO_SHARED_ENGINES"] = str(args.engines)
    os.environ.setdefault("HELLO_SHARED_IDLE_EXIT", "1")      # this run's server leaves as soon as its workers have (few processes fit on the card)
    spec_name = bench.BENCH_CONFIGS[args.config]["spec"]
    spec = ns.build(spec_name)
    model_path = os.path.join(tmp, spec_name + ".npz")
    loader.save_native(model_path, spec_name, weights.synth_state(spec, seed=1))
    ctx = mp.get_context("spawn")
    start_evt, q = ctx.Event(), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, args.calls, start_evt, q, model_path, args.shared, args.config)) for r in range(args.workers)]
    for p in procs:
        p.start()
    for _ in procs:
        assert q.get(timeout=600)[0] == "ready"
    t0 = time.perf_counter()
    start_evt.set()
    rates = sorted((q.get(timeout=900) for _ in procs), key=lambda x: x[1])
    wall = time.perf_counter() - t0
    for p in procs:
        p.join()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)         # the model file and the rendezvous directory of this run (the server unlinks its own files)
    if args.json:
        stats = max((r[3] for r in rates), key=lambda s: s["sites"]) if args.shared else None
        if stats is not None:
            args.engines = stats["engines"]
        print(json.dumps({"value": round(sum(r[2] for r in rates), 1), "unit": "sites/s", "workers": args.workers, "calls_per_worker": args.calls,
                          "by_wall_clock_of_slowest": round(args.workers * args.calls / wall, 1), "config": args.config,
                          "ms_per_call_mean": round(1e3 * args.workers / sum(r[2] for r in rates), 4),
                          "form": "loader.load(path, shared=True): one native server process" if args.shared else "an engine per worker",
                          "engines": args.engines if args.shared else args.workers, "server": stats}))
        return
    for _, rank, rate, _ in rates:
        print(f"  worker {rank}: {rate:8.0f} sites/s ({1e3 / rate:.3f} ms per call)")
    if args.shared:
        args.engines = max((r[3] for r in rates), key=lambda s: s["sites"])["engines"]          # what the server chose
    form = f"loader.load(path, shared=True): one server process, {args.engines} engine(s)" if args.shared else "an engine per worker"
    print(f"{args.workers} worker processes ({args.config}), one site per call each, {form}: {sum(r[2] for r in rates):,.0f} sites/s in aggregate "
          f"({args.workers * args.calls / wall:,.0f} by the wall clock of the slowest worker)")
    if args.shared:
        stats = max((r[3] for r in rates), key=lambda s: s["sites"])
        print(f"  server: {json.dumps(stats)}; mean sites per launch {stats['sites'] / max(stats['launches'], 1):.1f}")


if __name__ == "__main__":
    main()
