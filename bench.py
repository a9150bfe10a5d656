#!/usr/bin/env python3
"""Headline benchmark: candidate sites/s of the variant-scoring hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): Illumina 30x single-tech model
(moe_attention_config_single_tech_old_equivalent_weight_norm), synthetic sites from the seeded
generator, seeded synthetic weights.  One *step* = one forward of the hot path over one batch of
``--sites`` candidate sites already resident in HBM: uint8 pileups + CSR counts -> allele logits ->
genotype-pair posteriors, all on the GPU.  ``value`` = sites processed by all ranks / wall time of
exactly K steps (barrier + synchronize on both sides, max over ranks).  With N > 1 every rank scores
its own shard of sites (weak scaling, no data-path collective) and the per-rank logits are gathered
once to rank 0 over RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sites S] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0
PARITY_SITES = 96


def site_flops(spec, batch):
    """Algorithmic FLOPs of one batch (2 * MAC; dead site-level compressor excluded)."""
    from hello_amd import netspec as ns
    per_read = 2 * ns.macs(spec.nets["read_convolver0"], spec.window)
    per_allele = 2 * (ns.macs(spec.nets["compressor0"], 36) + ns.macs(spec.nets["xattn0"], 18))
    return per_read * batch.reads0.shape[0] + per_allele * batch.n_alleles, per_read, per_allele


def host_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def _per_site_worker(job):
    """One worker process of the reference's deployment form (call.py:26-30,111,215-221): torch on ONE
    thread, one site per call through the per-site wrapper.  -> (sites scored, seconds)."""
    seed, worker, budget_s = job
    import torch
    torch.set_num_threads(1)
    from hello_amd import netspec as ns, synth, weights
    from oracle import moe_oracle as mo
    spec = ns.build("single_tech")
    wrapper = mo.WrapperOracle(spec, weights.synth_state(spec, seed=seed), backend="torch")
    sample = synth.make_sites(48, seed=seed + 999 + worker, coverage=30)
    names = synth.allele_names(sample)
    aoff = np.concatenate([[0], np.cumsum(sample.alleles_per_site)])
    roff = np.concatenate([[0], np.cumsum(sample.reads_per_allele0)])

    def score(s):
        fd = {names[s][k]: (sample.reads0[roff[a]:roff[a + 1]].astype(np.float32), None)
              for k, a in enumerate(range(aoff[s], aoff[s + 1]))}
        wrapper(fd, sample.ref_onehot[s:s + 1].astype(np.float32))

    score(0)                                                           # warm
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        score(n % sample.n_sites)
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline(seed, budget_s=12.0):
    """The CPU oracle (torch-CPU conv back end = the reference's own third-party kernels) on a bounded sample
    of the same workload, in the reference's deployment form: one single-threaded worker process per usable
    host core, one site per call.  The batched all-threads form is reported beside it.  Must run before this
    process touches the GPU (it forks)."""
    import multiprocessing as mp
    import torch
    from hello_amd import netspec as ns, synth, weights
    from oracle import moe_oracle as mo
    cores = host_cores()
    with mp.get_context("fork").Pool(cores) as workers:
        res = workers.map(_per_site_worker, [(seed, w, budget_s) for w in range(cores)])
    rate = sum(n / dt for n, dt in res)
    done_sites = sum(n for n, _ in res)

    spec = ns.build("single_tech")
    torch.set_num_threads(cores)
    oracle = mo.Oracle(spec, weights.synth_state(spec, seed=seed), backend="torch")
    chunk = 64
    sample = synth.make_sites(chunk * 8, seed=seed + 999, coverage=30)
    mo.forward_batch(oracle, sample.site_slice(0, 8), chunk_sites=8)      # warm
    done, t0 = 0, time.perf_counter()
    while done < sample.n_sites and time.perf_counter() - t0 < budget_s / 2:
        mo.forward_batch(oracle, sample.site_slice(done, done + chunk), chunk_sites=chunk)
        done += chunk
    dt = time.perf_counter() - t0
    # the other half of BASELINE.json's metric: what the CPU path answers on a fixed sample, kept aside for
    # the comparison with the engine's answers on the same sites (``parity`` in the JSON line)
    check = synth.make_sites(PARITY_SITES, seed=seed + 4242, coverage=30)
    want_logits, _ = mo.forward_batch(mo.Oracle(spec, weights.synth_state(spec, seed=seed)), check, chunk_sites=1)
    aoff = np.concatenate([[0], np.cumsum(check.alleles_per_site)])
    probs = mo.sigmoid(want_logits[0])
    want_post = np.concatenate([mo.posteriors([probs[aoff[s]:aoff[s + 1]]] + [np.zeros(aoff[s + 1] - aoff[s], np.float32)] * 2,
                                              np.array([1, 0, 0], np.float32))[0] for s in range(check.n_sites)])
    cpu_baseline.reference_answers = (check, probs, want_post)
    return {"value": round(rate, 2), "unit": "sites/s", "cores": int(cores), "kind": "port",
            "sample": f"{done_sites} synthetic sites (cov 30) in {budget_s:.0f} s: {cores} single-threaded worker "
                      f"processes, one site per call through oracle/moe_oracle.py's per-site wrapper (the "
                      f"reference's deployment form, call.py:26-30,111), torch-CPU conv back end",
            "per_core": round(rate / cores, 2),
            "batched_all_threads": {"value": round(done / dt, 2), "unit": "sites/s", "cores": int(cores),
                                    "sample": f"{done} sites in chunks of {chunk}, one process, {dt:.1f} s"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sites", type=int, default=8192, help="candidate sites per step per GPU")
    ap.add_argument("--pool", type=int, default=2, help="distinct resident batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 256-sites-per-launch leg")
    ap.add_argument("--fused", choices=["full", "trunk", "none"], default="full",
                    help="read convolver: one fused kernel from the bytes / layer-by-layer stem + fused trunk / "
                         "layer by layer")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--op-times", action="store_true", help="print per-op device times to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CPU baseline first: it forks worker processes, which must happen before this process touches the GPU
    # (and is skipped under a profiler, whose preloaded library has initialised the GPU already)
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not profiled:
        try:
            cpu = cpu_baseline(args.seed)
        except Exception as exc:           # the GPU measurement must not be lost to a host-side hiccup (fork limits ...)
            print(f"cpu baseline failed: {exc!r}", file=sys.stderr)
            cpu = {"value": None, "unit": "sites/s", "cores": 0, "kind": "port", "sample": f"failed: {exc!r}"}

    import torch
    from hello_amd import netspec as ns, synth, weights
    from hello_amd.engine import Engine, n_pairs

    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    # HELLO_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("HELLO_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run (also at world size 1)
        import torch.distributed as dist
        # RCCL prints a version banner on stdout when its communicator comes up; stdout is reserved for the one
        # JSON line, so the process's fd 1 points at stderr until the first collective has completed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=args.seed)
    eng = Engine(spec, state, device=dev_index, fused={"full": True, "trunk": "trunk", "none": False}[args.fused])

    # resident pool of synthetic batches (every rank its own sites: shard = rank)
    pool = []
    for i in range(args.pool):
        b = synth.make_sites(args.sites, seed=1000 * (rank + 1) + i + args.seed, coverage=30)
        pool.append(dict(
            batch=b, reads=torch.from_numpy(b.reads0).to(dev), rpa=b.reads_per_allele0,
            aps=b.alleles_per_site, pairs=n_pairs(b.alleles_per_site)))
    max_a = max(p["batch"].n_alleles for p in pool)
    max_p = max(p["pairs"] for p in pool)
    if dist is not None:
        # ranks hold different random shards: agree on one row width so the final gather is a plain gather
        width = torch.tensor([max_a], dtype=torch.int64, device=dev)
        dist.all_reduce(width, op=dist.ReduceOp.MAX)
        max_a = int(width.item())
    # outputs of every timed step stay on the device until the single gather at the end
    out_logits = torch.zeros((args.steps, max_a), dtype=torch.float32, device=dev)
    out_post = torch.zeros((4, max_p), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(i, record=None):
        p = pool[i % len(pool)]
        a = p["batch"].n_alleles
        lg = out_logits[record if record is not None else 0, :a].view(1, a)
        eng.forward(p["reads"], p["rpa"], p["aps"], stream=stream,
                    out=(lg, None, out_post[:, :p["pairs"]]), posteriors=True)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    fence()
    eng.set_profiling(args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, record=i)
    gathered = None
    if dist is not None:
        # the one collective of the path: per-rank logits -> rank 0
        gathered = [torch.empty_like(out_logits) for _ in range(world)] if rank == 0 else None
        dist.gather(out_logits, gathered, dst=0)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    sites_local = sum(pool[i % len(pool)]["batch"].n_sites for i in range(args.steps))
    sites_total = sites_local * world
    value = sites_total / dt

    # ---- roofline of the dominant kernel, from HIP events recorded around every op of the timed steps
    op_rows, n_fw = eng.op_times_ms()
    eng.set_profiling(0)
    if args.op_times and rank == 0:
        for i, (k, n, ms) in enumerate(op_rows):
            print(f"  op {i:3d} {k:15s} {ms:9.4f} ms  {n}", file=sys.stderr)
        print(f"  sum of ops {sum(r[2] for r in op_rows):.3f} ms over {n_fw} forwards", file=sys.stderr)
    flops_step = np.mean([site_flops(spec, pool[i % len(pool)]["batch"])[0] for i in range(args.steps)])
    reads_step = np.mean([pool[i % len(pool)]["batch"].reads0.shape[0] for i in range(args.steps)])
    alleles_step = np.mean([pool[i % len(pool)]["batch"].n_alleles for i in range(args.steps)])
    rows_of = {0: reads_step, 1: 0.0, 2: alleles_step, 3: float(args.sites)}
    fused = eng.program.fused_read_convolver
    # group the per-op event times by the kernel that ran them; algorithmic FLOPs = 2 * MAC of the op
    kernels = {}
    for op, (k, n, ms) in zip(eng.program.ops, op_rows):
        kname = {"conv1d": "conv1d_mfma_kernel", "readconv_fused": "readconv_kernel"}.get(k, k + "_kernel")
        if k == "conv1d" and (op.flags & 32):
            kname = "conv1d_wino_kernel"
        ent = kernels.setdefault(kname, dict(ms=0.0, flops=0.0, exec=0.0, launches=0))
        ent["ms"] += ms
        rows = reads_step if k == "readconv_fused" else rows_of[op.domain]   # the trunk's MACs are per read
        ent["flops"] += 2.0 * op.macs_per_row * rows
        ent["exec"] += 2.0 * (op.exec_macs_per_row or op.macs_per_row) * rows
        ent["launches"] += 1
    dom_name = max(kernels, key=lambda kk: kernels[kk]["ms"])
    dom = kernels[dom_name]
    dom_ms = dom["ms"] / dom["launches"]                 # average launch duration
    dom_flops = dom["flops"] / dom["launches"]           # algorithmic FLOPs per launch
    achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    executed = dom["exec"] / dom["launches"] / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
        "launch_ms": round(dom_ms, 4), "flop_per_launch": float(dom_flops), "launches_per_step": dom["launches"],
        # `achieved` prices the ALGORITHMIC work (direct-form 2*MAC) against the FP32 MFMA peak; the
        # k3/s1 convolutions run in Winograd form (F(3,3): 5 instead of 9 fp32 contractions per 3 positions; F(2,3):
        # 4 instead of 6 per 2), so the matrix cores execute fewer FLOPs than that: their own rate and utilisation are
        "mfma_executed_tflops": round(executed, 3), "mfma_executed_frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
        "arithmetic": "fp32; k3/s1 convolutions in Winograd form (residual trunk and allele stage F(3,3), stem F(2,3))" if eng.program.winograd else "fp32, direct form",
        "kernels_ms_per_step": {kk: round(v["ms"], 4) for kk, v in sorted(kernels.items(), key=lambda x: -x[1]["ms"])},
        "whole_step_frac": round(flops_step / (dt / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
        "hbm_algorithmic_gbs": round((900.0 * reads_step) / (dt / args.steps) / 1e9, 3),
    }

    # BASELINE.json's config 2 names launches of 256 sites: such a launch cannot fill the chip alone, so small
    # batches are alternated over four engines (own scratch and stream each); reported beside the headline value
    small = None
    if world == 1 and not args.no_small_batch:
        try:
            small_engines = [eng] + [Engine(spec, state, device=dev_index) for _ in range(3)]
            streams = [torch.cuda.Stream(dev) for _ in small_engines]
            sb = synth.make_sites(256, seed=args.seed + 77, coverage=30)
            sreads = torch.from_numpy(sb.reads0).to(dev)
            torch.cuda.synchronize(dev)

            def small_step(i):
                k = i % len(small_engines)
                with torch.cuda.stream(streams[k]):
                    small_engines[k].forward(sreads, sb.reads_per_allele0, sb.alleles_per_site,
                                             stream=streams[k].cuda_stream, posteriors=True)
            for i in range(8):
                small_step(i)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            n_small = 400
            for i in range(n_small):
                small_step(i)
            torch.cuda.synchronize(dev)
            dt_small = time.perf_counter() - t1
            small = {"sites_per_launch": 256, "engines": len(small_engines), "launches": n_small,
                     "value": round(256 * n_small / dt_small, 1), "unit": "sites/s",
                     "ms_per_launch": round(1e3 * dt_small / n_small, 4)}
            for e in small_engines[1:]:
                e.close()
        except Exception as exc:
            print(f"small-batch leg failed: {exc!r}", file=sys.stderr)

    parity = None
    if cpu is not None and getattr(cpu_baseline, "reference_answers", None) is not None:
        check, want_probs, want_post = cpu_baseline.reference_answers
        got_logits, _, got_post = eng.forward_batch(check, posteriors=True)
        got_probs = 1.0 / (1.0 + np.exp(-got_logits[0].astype(np.float64)))
        parity = {"max_abs_delta_allele_probability": float(np.abs(got_probs - want_probs).max()),
                  "max_abs_delta_pair_posterior": float(np.abs(got_post[0] - want_post).max()),
                  "tolerance": 1e-4, "sites": int(check.n_sites), "alleles": int(check.n_alleles),
                  "against": "oracle/moe_oracle.py (NumPy back end), one site per call"}

    if rank == 0:
        b0 = pool[0]["batch"]
        line = {
            "metric": "candidate sites/sec (whole node)", "value": round(value, 1), "unit": "sites/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Illumina 30x single-tech model (moe_attention single_tech weight_norm), "
                                   "synthetic pileups cov 30, seeded synthetic weights",
                       "sites_per_step_per_gpu": args.sites, "sites_total": sites_total,
                       "reads_per_site": round(b0.reads0.shape[0] / b0.n_sites, 2),
                       "alleles_per_site": round(b0.n_alleles / b0.n_sites, 3),
                       "window": 150, "channels": 6, "parallelism": f"site-sharded dp{world}",
                       "fused_read_convolver": bool(fused), "outputs": "logits + genotype-pair posteriors"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": parity,
            "small_batch": small,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
