#!/usr/bin/env python3
"""Headline benchmark: candidate sites/s of the variant-scoring hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): Illumina 30x single-tech model
(moe_attention_config_single_tech_old_equivalent_weight_norm), >= 1 M synthetic candidate sites from the
seeded generator, seeded synthetic weights.  ``--config {C2,C3,C4,C5,hybrid_full}`` makes any other BASELINE.json
configuration the timed region (same path, any N); the default (C2) run also reports C3, C4, C5 and hybrid_full in a
``configs`` block: host-to-host rate at ``--sites`` sites per launch over ``--config-launches`` launches, the read convolver's
roofline fraction from the engine's HIP events, and max |delta| against the oracle's per-site answers on 96 check sites.

``value`` is SURVEY.md 8d metric (1): host-resident uint8 pileups + CSR counts -> host-resident allele logits
AND genotype-pair posteriors, PCIe both ways included, through the product's own feed
(hello_amd.shard.partition_sites -> hello_amd.pipeline.HostPipeline -> hello_amd.engine.Engine).  One *step* =
one pass of the hot path over one batch of ``--launches-per-step`` x ``--sites`` candidate sites per GPU
(10 x 8 192 = 81 920 by default, scored in launches of ``--sites``; the launches cycle a pinned pool of
``--pool`` distinct synthetic batches).  Exactly K steps are timed between barrier + synchronize fences, with
every result harvested to host memory before the closing fence.  The device-resident rate (inputs already in
HBM, outputs left there) is reported beside it as ``device_resident``: the pipeline hides the PCIe feed behind
the kernels, so the two agree within a few percent.

``python bench.py --gpus N`` from a plain shell starts its own N ranks (a CHILD ``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1``, before this process has touched a GPU; never an exec) and relays rank 0's one JSON line;
under a launcher (RANK / WORLD_SIZE set: the driver's form) the process is a rank.  The CPU baseline is timed on rank 0 at every N,
before the rendezvous (the other ranks wait idle in it), so the line carries ``cpu_baseline`` at N > 1 too.

With N > 1 (one process per GPU under torch.distributed.run) the global step batch is N times as large
(``--scaling weak``, the default) or the same ``--launches-per-step`` launches cut N ways (``--scaling strong``:
the fixed 1.64 M-site stream of the N = 1 run); every rank computes the same read-balanced site partition from
the shared counts, pins and feeds ONLY its range through its own pipeline (host threads on CPUs near its GPU) and
keeps its logits (and, for ensemble models, its per-site meta weights) resident; ONE RCCL gather at the end of the run
brings every rank's results to rank 0, inside the timed region.

At N > 1 the line also carries the run's audit trail, collected with one ``all_gather_object`` AFTER the closing fence:
``ranks`` (per rank: device index, PCI address, UUID, sites, reads, own seconds, launches, pinned bytes, CPUs, ``own_checksum`` of the
columns it produced), ``distinct_devices``, ``backend``, ``gather_ms`` (two events around the one gather), ``slowest_rank``,
``balance`` = min / max reads per rank, ``gather_verified_ranks`` = the number of ranks whose own checksum equals the checksum of
what rank 0 received for them (the gather proves itself; the run fails when it is not N); and with ``--scaling both`` (the
default) ``strong_scaling``: a second timed region of the same K steps over the N = 1 stream cut N ways.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5|hybrid_full] [--sites S] [--launches-per-step B]
                    [--scaling weak|strong|both]
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


# BASELINE.json's configurations (SURVEY.md 8d "Concrete configs"): the model + the synthetic generator's arguments, exactly as
# tests/test_gpu_fullsize.py builds them.  C2 is the headline (BASELINE.json configs[1], the configuration `metric` is quoted on);
# hybrid_full is not a BASELINE config: it is here because its `meta` mixing weights cross the gather (three experts + meta).
BENCH_CONFIGS = {
    "C2": dict(spec="single_tech", sites=dict(coverage=30),
               label="Illumina 30x single-tech model (moe_attention single_tech weight_norm), synthetic pileups cov 30"),
    "C3": dict(spec="single_tech", sites=dict(coverage=(8, 52), tech="pacbio", max_reads=128),
               label="PacBio HiFi single-tech model, long-window pileups, coverage U{8..52}, <= 128 reads per site"),
    "C4": dict(spec="hybrid_no_ensemble", sites=dict(coverage=30, hybrid_coverage=15),
               label="hybrid Illumina 30x + PacBio 15x no-ensemble model (moe_attention full_hybrid weight_norm no_ensemble: two read "
                     "convolvers, combiners, one expert)"),
    "C5": dict(spec="single_tech_hp", sites=dict(coverage=(20, 80), channels=7, tech="pacbio"),
               label="PacBio haplotagged model (7 channels: read_convolver_with_hp_channel), mixed coverage U{20..80}, ragged CSR packing"),
    "hybrid_full": dict(spec="hybrid_full", sites=dict(coverage=30, hybrid_coverage=15),
                        label="full hybrid Illumina 30x + PacBio 15x model (three experts + meta mixing weights)"),
}
SECONDARY_CONFIGS = ("C3", "C4", "C5", "hybrid_full")


def make_config_sites(name, n_sites, seed):
    from hello_amd import synth
    return synth.make_sites(n_sites, seed=seed, **BENCH_CONFIGS[name]["sites"])


def _rows_of(op, batch):
    """Rows one op of a compiled program walks over, for a batch (reads of the op's technology / alleles / sites)."""
    from hello_amd import compiler
    if op.kind == compiler.OP_READCONV_FUSED:
        return int(batch.reads0.shape[0]) if op.seg == compiler.SEG_R0A else (0 if batch.reads1 is None else int(batch.reads1.shape[0]))
    return {compiler.ROWS_READS0: int(batch.reads0.shape[0]), compiler.ROWS_READS1: 0 if batch.reads1 is None else int(batch.reads1.shape[0]),
            compiler.ROWS_ALLELES: batch.n_alleles, compiler.ROWS_SITES: batch.n_sites}[op.domain]


def program_flops(program, batch):
    """-> (algorithmic FLOPs of one batch: 2 * direct-form MAC of every op the engine runs -- SURVEY 8d's per-read / per-allele /
    per-site figures, dead site-level compressor excluded --, FLOPs the matrix instructions execute: Winograd forms counted as run)."""
    alg = sum(2.0 * op.macs_per_row * _rows_of(op, batch) for op in program.ops)
    exe = sum(2.0 * (op.exec_macs_per_row or op.macs_per_row) * _rows_of(op, batch) for op in program.ops)
    return alg, exe


def batch_input_bytes(batch):
    """Algorithmic input bytes of one batch: the uint8 pileups of both technologies (R * L * C)."""
    n = int(np.prod(batch.reads0.shape))
    return n + (0 if batch.reads1 is None else int(np.prod(batch.reads1.shape)))


def feature_dicts(batch):
    """A SiteBatch as the per-site call's arguments: [(featureDict {allele: (float [R, L, C], float [R', L, C] | None)}, segment)]
    (caller_calling.py:631-649)."""
    from hello_amd import synth
    names = synth.allele_names(batch)
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    roff0 = np.concatenate([[0], np.cumsum(batch.reads_per_allele0)])
    roff1 = None if batch.reads1 is None else np.concatenate([[0], np.cumsum(batch.reads_per_allele1)])
    out = []
    for s in range(batch.n_sites):
        fd = {}
        for k, a in enumerate(range(aoff[s], aoff[s + 1])):
            second = None if roff1 is None else batch.reads1[roff1[a]:roff1[a + 1]].astype(np.float32)
            fd[names[s][k]] = (batch.reads0[roff0[a]:roff0[a + 1]].astype(np.float32), second)
        out.append((fd, batch.ref_onehot[s:s + 1].astype(np.float32)))
    return out


def _oracle_answers_worker(job):
    """One worker of the reference-answer pool: sites [lo, hi) of a configuration's check batch through oracle/moe_oracle.py, ONE
    SITE PER CALL (the reference's per-site form, MixtureOfExpertsAdvanced.py:520-589).  -> (lo, probabilities [E, A_part],
    meta [S_part, 3] | None, posteriors [4, P_part])."""
    name, seed, n_sites, lo, hi = job
    import torch
    torch.set_num_threads(1)
    try:                                    # NumPy's BLAS threads too: one worker per core, not cores x cores threads
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from hello_amd import netspec as ns, weights
    from oracle import moe_oracle as mo
    spec = ns.build(BENCH_CONFIGS[name]["spec"])
    oracle = mo.Oracle(spec, weights.synth_state(spec, seed=seed))            # NumPy back end: the independent restatement
    sub = make_config_sites(name, n_sites, seed + 4242).site_slice(lo, hi)
    logits, meta = mo.forward_batch(oracle, sub, chunk_sites=1)
    probs = mo.sigmoid(logits)
    aoff = np.concatenate([[0], np.cumsum(sub.alleles_per_site)])
    post = []
    for s in range(sub.n_sites):
        p = [probs[e, aoff[s]:aoff[s + 1]] for e in range(probs.shape[0])]
        if len(p) == 1:                     # single-expert models: experts [e0, 0, 0], meta [1, 0, 0]  (:530-538)
            p, m = p + [np.zeros_like(p[0])] * 2, np.array([1, 0, 0], np.float32)
        else:
            m = meta[s]
        post.append(np.stack(mo.posteriors(p, m)))
    return lo, probs, meta, np.concatenate(post, axis=1)


def oracle_answers(names, seed, n_sites=96):
    """What the CPU oracle answers on ``n_sites`` seeded check sites of each configuration (the other half of BASELINE.json's
    metric: max |delta| vs the CPU reference).  A pool of single-threaded workers forked BEFORE this process touches the GPU.
    -> {name: (check batch, probabilities [E, A], meta | None, posteriors [4, P])}."""
    import multiprocessing as mp
    cores = host_cores()
    cut = max(1, min(cores, 8))
    jobs = []
    for name in names:
        edges = np.linspace(0, n_sites, cut + 1).astype(int)
        jobs += [(name, seed, n_sites, int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]
    with mp.get_context("fork").Pool(min(cores, len(jobs))) as workers:
        parts = workers.map(_oracle_answers_worker, jobs)
    out = {}
    for name in names:
        mine = sorted((p for j, p in zip(jobs, parts) if j[0] == name), key=lambda p: p[0])
        meta = None if mine[0][2] is None else np.concatenate([p[2] for p in mine], axis=0)
        out[name] = (make_config_sites(name, n_sites, seed + 4242), np.concatenate([p[1] for p in mine], axis=1), meta,
                     np.concatenate([p[3] for p in mine], axis=1))
    return out


def parity_of(engine, answers):
    """The engine's answers on a configuration's check sites against the oracle's: max |delta| of sigmoid(logit) over every expert,
    of the meta weights, and of every genotype-pair posterior row the model produces (tolerance 1e-4, BASELINE.json north_star)."""
    check, want_probs, want_meta, want_post = answers
    got_logits, got_meta, got_post = engine.forward_batch(check, posteriors=True)
    got_probs = 1.0 / (1.0 + np.exp(-got_logits.astype(np.float64)))
    rows = 4 if engine.n_experts > 1 else 1          # single-expert models: the mixture row IS expert 0's
    out = {"max_abs_delta_allele_probability": float(np.abs(got_probs - want_probs).max()),
           "max_abs_delta_pair_posterior": float(np.abs(got_post[:rows] - want_post[:rows]).max()),
           "tolerance": 1e-4, "sites": int(check.n_sites), "alleles": int(check.n_alleles), "experts": int(engine.n_experts),
           "against": "oracle/moe_oracle.py (NumPy back end), one site per call"}
    if want_meta is not None:
        out["max_abs_delta_meta"] = float(np.abs(got_meta - want_meta).max())
    # "identical calls": the call of a site is its most probable genotype pair of the mixture row (caller_calling.py:698-712) and
    # QUAL = -10 log10(1 - min(p, 1 - 1e-8)) (vcfFromContigs.py:215-220): the same pair on every check site, and how far QUAL moves
    a = np.asarray(check.alleles_per_site, np.int64)
    p_off = np.concatenate([[0], np.cumsum(a * (a + 1) // 2)])
    best = lambda post: np.array([int(np.argmax(post[0, p_off[s]:p_off[s + 1]])) for s in range(check.n_sites)])     # noqa: E731
    top = lambda post: np.array([float(post[0, p_off[s]:p_off[s + 1]].max()) for s in range(check.n_sites)])         # noqa: E731
    qual = lambda p: -10.0 * np.log10(1.0 - np.minimum(p, 1.0 - 1e-8))                                               # noqa: E731
    out["sites_with_identical_call"] = int((best(got_post) == best(want_post)).sum())
    out["qual_max_abs_delta"] = float(np.abs(qual(top(got_post.astype(np.float64))) - qual(top(want_post.astype(np.float64)))).max())
    out["within_tolerance"] = bool(max(v for k, v in out.items() if k.startswith("max_abs_delta")) <= 1e-4)
    return out


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
    config, seed, worker, budget_s = job
    import torch
    torch.set_num_threads(1)
    from hello_amd import netspec as ns, weights
    from oracle import moe_oracle as mo
    spec = ns.build(BENCH_CONFIGS[config]["spec"])
    wrapper = mo.WrapperOracle(spec, weights.synth_state(spec, seed=seed), backend="torch")
    calls = feature_dicts(make_config_sites(config, 48, seed + 999 + worker))
    wrapper(*calls[0])                                                 # warm
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        wrapper(*calls[n % len(calls)])
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline(seed, budget_s=12.0, config="C2"):
    """The CPU oracle (torch-CPU conv back end = the reference's own third-party kernels) on a bounded sample
    of the same workload, in the reference's deployment form: one single-threaded worker process per usable
    host core, one site per call.  The batched all-threads form is reported beside it.  Must run before this
    process touches the GPU (it forks)."""
    import multiprocessing as mp
    import torch
    from hello_amd import netspec as ns, weights
    from oracle import moe_oracle as mo
    cores = host_cores()
    with mp.get_context("fork").Pool(cores) as workers:
        res = workers.map(_per_site_worker, [(config, seed, w, budget_s) for w in range(cores)])
    rate = sum(n / dt for n, dt in res)
    done_sites = sum(n for n, _ in res)

    spec = ns.build(BENCH_CONFIGS[config]["spec"])
    torch.set_num_threads(cores)
    oracle = mo.Oracle(spec, weights.synth_state(spec, seed=seed), backend="torch")
    chunk = 64
    sample = make_config_sites(config, chunk * 8, seed + 999)
    mo.forward_batch(oracle, sample.site_slice(0, 8), chunk_sites=8)      # warm
    done, t0 = 0, time.perf_counter()
    while done < sample.n_sites and time.perf_counter() - t0 < budget_s / 2:
        mo.forward_batch(oracle, sample.site_slice(done, done + chunk), chunk_sites=chunk)
        done += chunk
    dt = time.perf_counter() - t0
    return {"value": round(rate, 2), "unit": "sites/s", "cores": int(cores), "kind": "port", "config": config,
            "sample": f"{done_sites} synthetic sites of {config} in {budget_s:.0f} s: {cores} single-threaded worker "
                      f"processes, one site per call through oracle/moe_oracle.py's per-site wrapper (the "
                      f"reference's deployment form, call.py:26-30,111), torch-CPU conv back end",
            "per_core": round(rate / cores, 2),
            "batched_all_threads": {"value": round(done / dt, 2), "unit": "sites/s", "cores": int(cores),
                                    "sample": f"{done} sites in chunks of {chunk}, one process, {dt:.1f} s"}}


def shard_cpu_ranges(cpus):
    """[0, 1, 2, 5] -> "0-2,5" (a rank's CPU list, compact)."""
    out, run = [], []
    for c in sorted(cpus):
        if run and c == run[-1] + 1:
            run.append(c)
        else:
            if run:
                out.append(run)
            run = [c]
    if run:
        out.append(run)
    return ",".join(str(r[0]) if len(r) == 1 else f"{r[0]}-{r[-1]}" for r in out)


def rank_pieces(pool_counts, launches_total, rank, world):
    """The global step batch is ``launches_total`` pool batches back to back (cycling the pool).  Every rank
    computes the same read-balanced partition of its sites (hello_amd.shard.partition_sites) from the shared
    counts and gets its contiguous range as a list of (pool index, site lo, site hi) pieces -- whole pool
    batches except, for N > 1, where a cut falls inside one.  -> (pieces, sizes of every rank's range)."""
    from hello_amd import shard
    seq = [i % len(pool_counts) for i in range(launches_total)]
    reads = np.concatenate([pool_counts[k]["reads_per_site"] for k in seq])
    aps = np.concatenate([pool_counts[k]["alleles_per_site"] for k in seq])
    ranges = shard.partition_sites(reads, world)
    sizes = shard.shard_sizes(aps, ranges)
    lo, hi = ranges[rank]
    pieces, base = [], 0
    for k in seq:
        n = pool_counts[k]["alleles_per_site"].shape[0]
        a, b = max(lo, base), min(hi, base + n)
        if b > a:
            pieces.append((k, a - base, b - base))
        base += n
    return pieces, sizes


def launch_ranks(gpus, argv):
    """``python bench.py --gpus N`` from a plain command line: start the N ranks as ONE CHILD process tree under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and wait for it.  Called before this process has
    imported torch or touched a GPU; it never replaces itself (no exec), it only relays: the child inherits stdout, so rank
    0's one JSON line is this command's one line, and the child's exit code is this command's.  The same pattern as
    hello_amd/call.py's ``_launch_ranks`` (the reference's counterpart is its worker pool, call.py:111,215-221)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    rest, skip = [], False
    for a in argv:                                   # the child's own --gpus is appended below
        if skip:
            skip = False
        elif a == "--gpus":
            skip = True
        elif not a.startswith("--gpus="):
            rest.append(a)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(gpus)] + rest
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), HELLO_BENCH_SELF_LAUNCHED="1")
    print(f"bench: starting {gpus} ranks as a child: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def lib_sha256():
    """sha256 of the HIP library this run LOADED -- the file hello_amd.engine opened (HELLO_LIB when it is set: the kernel A/B case
    this guard exists for), never a path assumed here: ties `roofline.traffic` to the binary the committed PMC passes profiled
    (profiles/hbm_traffic.json records the same hash, tools/profile_round.sh).  None when the file cannot be read: the traffic is
    then reported as stale instead of the measurement being lost after the run."""
    import hashlib
    try:
        from hello_amd import engine
        h = hashlib.sha256()
        with open(engine._LIB_PATH, "rb") as fh:
            for chunk in iter(lambda: fh.read(1 << 20), b""):
                h.update(chunk)
        return h.hexdigest()
    except Exception as exc:
        print(f"bench: could not hash the loaded library: {exc!r}", file=sys.stderr)
        return None


def committed_traffic(path, loaded_sha):
    """-> (traffic, forward_traffic, provenance).  The HBM bytes of profiles/hbm_traffic.json are reported only when that
    file says which library it profiled AND it is the library this run loaded; otherwise both are None and
    ``provenance["traffic_stale"]`` is True (a changed kernel must not carry the previous build's bytes)."""
    prov = {"lib_sha256": loaded_sha, "profiled_lib_sha256": None, "profiled_commit": None, "traffic_stale": None,
            "source": "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950-corrected: "
                      "2 x FETCH + WRITE; tools/profile_round.sh)"}
    if not os.path.exists(path):
        return None, None, prov
    try:
        tjson = json.load(open(path))
    except Exception:
        return None, None, prov
    prov["profiled_lib_sha256"] = tjson.get("lib_sha256")
    prov["profiled_commit"] = tjson.get("commit")
    prov["traffic_stale"] = not (loaded_sha is not None and tjson.get("lib_sha256") == loaded_sha)
    if prov["traffic_stale"]:
        return None, None, prov
    forward = None
    if tjson.get("bytes_per_forward") is not None:
        forward = {"bytes": tjson["bytes_per_forward"], "algorithmic_bytes": tjson.get("algorithmic_bytes_per_forward"),
                   "by_kernel": tjson.get("bytes_per_forward_by_kernel"), "source": prov["source"]}
    return tjson.get("bytes_per_launch"), forward, prov


def checksum(columns):
    """What a rank says about the float32 columns it produced, and what rank 0 says about the columns it received for that rank: the
    two must be EQUAL (same bytes -> same crc32; the sums are float64 over the same values in the same order)."""
    import zlib
    x = np.ascontiguousarray(columns, dtype=np.float32)
    flat = x.reshape(-1)
    return {"n": int(flat.size), "crc32": int(zlib.crc32(flat.tobytes())), "sum": float(flat.sum(dtype=np.float64)),
            "abs_sum": float(np.abs(flat).sum(dtype=np.float64)), "first": float(flat[0]) if flat.size else None,
            "last": float(flat[-1]) if flat.size else None}


def to_device(batch, dev):
    """A SiteBatch whose pileups (and reference one-hot) are resident on ``dev``."""
    import torch
    from hello_amd.synth import SiteBatch
    t = lambda x: None if x is None else (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(dev)   # noqa: E731
    return SiteBatch(t(batch.reads0), batch.reads_per_allele0, batch.alleles_per_site, t(batch.ref_onehot), t(batch.reads1),
                     batch.reads_per_allele1)


def device_outputs(eng, batch, dev):
    """Preallocated (logits, meta | None, posteriors) device tensors of one batch."""
    import torch
    from hello_amd.engine import n_pairs
    f32 = dict(dtype=torch.float32, device=dev)
    return (torch.empty((eng.n_experts, batch.n_alleles), **f32), torch.empty((batch.n_sites, 3), **f32) if eng.has_meta else None,
            torch.empty((4, n_pairs(batch.alleles_per_site)), **f32))


def readconv_roofline(eng, op_rows, batches):
    """The dominant kernel of a forward, priced from the engine's per-op HIP events: every fused read-convolver op of the program (one
    per read technology) -- or, in a program without one, the slowest op.  -> dict(kernel, launch_ms = mean device ms per forward
    summed over those ops, algorithmic / executed FLOPs per forward, ops)."""
    from hello_amd import compiler
    ops = eng.program.ops
    index = [i for i, o in enumerate(ops) if o.kind == compiler.OP_READCONV_FUSED]
    if not index:
        index = [max(range(len(op_rows)), key=lambda i: op_rows[i][2])]
    ms = sum(op_rows[i][2] for i in index)
    rows = lambda o: float(np.mean([_rows_of(o, b) for b in batches]))                                          # noqa: E731
    alg = sum(2.0 * ops[i].macs_per_row * rows(ops[i]) for i in index)
    exe = sum(2.0 * (ops[i].exec_macs_per_row or ops[i].macs_per_row) * rows(ops[i]) for i in index)
    fusedk = ops[index[0]].kind == compiler.OP_READCONV_FUSED
    return dict(kernel="readconv_kernel" if fusedk else op_rows[index[0]][0], fused=fusedk, launch_ms=ms, flops=alg, exec_flops=exe,
                n_ops=len(index), rows=sum(rows(ops[i]) for i in index))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(BENCH_CONFIGS), default="C2",
                    help="which BASELINE.json configuration the timed region runs (default C2 = configs[1], the one `metric` is "
                         "quoted on); the default run also reports C3, C4, C5 and hybrid_full in its `configs` block")
    ap.add_argument("--sites", type=int, default=8192, help="candidate sites per engine launch")
    ap.add_argument("--launches-per-step", type=int, default=10,
                    help="launches of --sites sites that make one step's batch on one GPU (10 x 8192 = 81 920 sites: "
                         "20 steps stream 1.64 M sites in ~3.4 s)")
    ap.add_argument("--pool", type=int, default=3, help="distinct pinned synthetic batches the launches cycle through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of the CPU baseline's per-site leg (its batched leg takes half)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the device-resident, latency, small-batch, parity and other-configuration legs (headline only)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (the other BASELINE configurations)")
    ap.add_argument("--no-shared-leg", action="store_true", help="skip the worker-pool leg (`per_site_shared`: a child process tree)")
    ap.add_argument("--config-launches", type=int, default=40, help="timed launches of each configuration of the `configs` block")
    ap.add_argument("--fused", choices=["full", "trunk", "none"], default="full",
                    help="read convolver: one fused kernel from the bytes / layer-by-layer stem + fused trunk / "
                         "layer by layer")
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="both",
                    help="N > 1: weak = every GPU gets --launches-per-step launches per step (global batch N times as "
                         "large); strong = the N = 1 stream (--launches-per-step launches per step in total) cut N ways; "
                         "both (default) = `value` is the weak run and a second timed region of the same K steps reports the "
                         "strong one as `strong_scaling` (at N = 1 the two coincide and only one region runs)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--op-times", action="store_true", help="print per-op device times to stderr")
    args = ap.parse_args()
    cfg = BENCH_CONFIGS[args.config]

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        # a plain command line asking for N GPUs: the ranks are a child process tree, this process only waits for it
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # CPU work first, on rank 0 at every N: the baseline and the oracle's reference answers fork worker processes, which must
    # happen before this process touches the GPU (and is skipped under a profiler, whose preloaded library has initialised the
    # GPU already).  At N > 1 the other ranks are blocked in the rendezvous below while it runs (no engine, no feeder threads
    # yet), so it has the host to itself exactly as at N = 1 and is over before any timed region starts.
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    secondary = world == 1 and not args.no_secondary
    other_configs = [c for c in SECONDARY_CONFIGS if c != args.config] if (secondary and args.config == "C2" and not args.no_configs) else []
    cpu, answers = None, {}
    if rank == 0 and not profiled:
        if not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(args.seed, budget_s=args.cpu_budget, config=args.config)
                if world > 1:
                    cpu["when"] = f"on rank 0 before the {world}-rank rendezvous (the other ranks wait idle in it)"
            except Exception as exc:           # the GPU measurement must not be lost to a host-side hiccup (fork limits ...)
                print(f"cpu baseline failed: {exc!r}", file=sys.stderr)
                cpu = {"value": None, "unit": "sites/s", "cores": 0, "kind": "port", "sample": f"failed: {exc!r}"}
        if secondary:
            try:
                t_or = time.perf_counter()
                answers = oracle_answers([args.config] + other_configs, args.seed, n_sites=PARITY_SITES)
                print(f"bench: oracle answers on {PARITY_SITES} check sites of {', '.join(answers)} in {time.perf_counter() - t_or:.1f} s",
                      file=sys.stderr)
            except Exception as exc:
                print(f"oracle reference answers failed: {exc!r}", file=sys.stderr)

    import torch
    from hello_amd import netspec as ns, shard, weights
    from hello_amd.engine import Engine
    from hello_amd.pipeline import HostPipeline, pin_batch

    if world != args.gpus:
        print(f"bench: --gpus {args.gpus} under a launcher of {world} rank(s): the launcher's world size is what runs", file=sys.stderr)
        args.gpus = world
    # HELLO_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("HELLO_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    # host threads of this rank (pinned staging, CSR building) on CPUs near its GPU, before the pinned pool exists
    cpus = shard.pin_rank(local_rank, local_world, dev_index) if world > 1 else sorted(os.sched_getaffinity(0))
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

    spec = ns.build(cfg["spec"])
    state = weights.synth_state(spec, seed=args.seed)
    # the headline is exact fp32 and says so from the engine's own record of what it computes in (never from a literal)
    eng = Engine(spec, state, device=dev_index, fused={"full": True, "trunk": "trunk", "none": False}[args.fused], arithmetic="fp32")
    if eng.program.arithmetic != "fp32":
        raise SystemExit(f"the headline engine computes in {eng.program.arithmetic!r}: `value` / dtype 'f32' are exact fp32 only")

    # ---- the pinned host pool (the same seeded batches on every rank: one global site stream) ---------------
    pool = [make_config_sites(args.config, args.sites, 1000 + i + args.seed) for i in range(args.pool)]
    counts = [dict(reads_per_site=shard.reads_per_site(b), alleles_per_site=b.alleles_per_site) for b in pool]
    pipe = HostPipeline(eng, depth=2, posteriors=True)
    pinned_cache = {}

    def pinned(k, lo, hi):
        """A rank pins what it feeds and nothing else: whole pool batches once each, a cut batch only as its slice."""
        key = (k, lo, hi)
        if key not in pinned_cache:
            pinned_cache[key] = pin_batch(pool[k] if (lo, hi) == (0, pool[k].n_sites) else pool[k].site_slice(lo, hi))
        return pinned_cache[key]

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize(dev)

    identity = shard.device_identity(dev_index)

    def n_reads(b):
        return int(b.reads0.shape[0]) + (0 if b.reads1 is None else int(b.reads1.shape[0]))

    def timed_region(mode, profile):
        """W untimed + exactly K timed steps of this rank's share of the global step batch under ``mode`` (weak | strong),
        fenced on both sides, results harvested to host and -- at N > 1 -- gathered to rank 0 inside the region.
        -> dict: the run's aggregate numbers (same on every rank) + this rank's own report."""
        global_launches = args.launches_per_step * (world if mode == "weak" else 1)
        pieces, sizes = rank_pieces(counts, global_launches, rank, world)
        piece_batches = [pinned(k, lo, hi) for k, lo, hi in pieces]
        step_sites = sum(hi - lo for _, lo, hi in pieces)
        step_alleles = sizes[rank][1]
        step_reads = sum(n_reads(b) for b in piece_batches)
        assert step_sites == sizes[rank][0] and step_alleles == sum(int(b.n_alleles) for b in piece_batches)
        # a rank's logits (and meta weights) of the whole run stay resident for the single gather at the end
        sink = sink_meta = None
        if dist is not None:
            sink = torch.zeros((eng.n_experts, max(args.steps, 1) * max(step_alleles, 1)), dtype=torch.float32, device=dev)
            if eng.has_meta:
                sink_meta = torch.zeros((max(args.steps, 1) * max(step_sites, 1), 3), dtype=torch.float32, device=dev)

        def run_steps(n_steps, keep):
            """n_steps passes over this rank's pieces through the pipeline; every result is harvested to host
            memory before this returns.  -> number of launches harvested."""
            harvested, col, row = 0, 0, 0
            for s in range(n_steps):
                for p, b in enumerate(piece_batches):
                    snk = (sink, sink_meta, col, row) if (sink is not None and keep is not None) else None
                    for item in pipe.submit(b, tag=(s, p), sink=snk):
                        harvested += 1
                        if keep is not None:
                            keep.append(item)
                    col += int(b.n_alleles)
                    row += int(b.n_sites)
            for item in pipe.flush():
                harvested += 1
                if keep is not None:
                    keep.append(item)
            return harvested

        run_steps(args.warmup, None)
        fence()
        n_launches = args.steps * len(piece_batches)
        if profile:
            eng.set_profiling(min(max(n_launches, 1), 4096), only="readconv_fused")     # two events per launch: the dominant kernel
        results = []
        t0 = time.perf_counter()
        harvested = run_steps(args.steps, results)
        t_scored = time.perf_counter()
        gathered, gathered_meta, gather_ms, gather_host_ms = None, None, None, None
        if dist is not None:
            # the one collective of the path: every rank's logits (+ meta) of the run -> rank 0, then to its host.
            # Timed with two events on the current stream (RCCL orders its work with it) and with the host clock.
            run_sizes = [(s * args.steps, a * args.steps) for s, a in sizes]
            ev0 = ev1 = None
            if backend == "nccl":
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            g0 = time.perf_counter()
            own_sites, own_alleles = run_sizes[rank]
            gathered, gathered_meta = shard.gather_results(sink[:, :own_alleles], None if sink_meta is None else sink_meta[:own_sites],
                                                           run_sizes, eng.n_experts, eng.has_meta, dst=0)
            if ev1 is not None:
                ev1.record()
            if gathered is not None:
                gathered = gathered.cpu()
                gathered_meta = None if gathered_meta is None else gathered_meta.cpu()
            if ev1 is not None:
                ev1.synchronize()
                gather_ms = float(ev0.elapsed_time(ev1))
            gather_host_ms = 1e3 * (time.perf_counter() - g0)
        own_dt = time.perf_counter() - t0                # this rank's clock before the closing fence (who was slowest)
        fence()
        dt = time.perf_counter() - t0
        assert harvested == n_launches, (harvested, n_launches)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        sites_total = sum(s for s, _ in sizes) * args.steps
        report = dict(rank=rank, local_rank=local_rank, host=os.uname().nodename, pid=os.getpid(), **identity,
                      sites=int(step_sites * args.steps), alleles=int(step_alleles * args.steps), reads=int(step_reads * args.steps),
                      launches=int(n_launches), timed_seconds=round(own_dt, 6), scored_seconds=round(t_scored - t0, 6),
                      gather_ms=None if gather_ms is None else round(gather_ms, 4),
                      gather_host_ms=None if gather_host_ms is None else round(gather_host_ms, 4),
                      pinned_input_bytes=int(sum(batch_input_bytes(b) for b in {id(b): b for b in piece_batches}.values())),
                      cpus_pinned=len(cpus), cpu_list=shard_cpu_ranges(cpus))
        return dict(mode=mode, dt=dt, value=sites_total / dt, sites_total=sites_total, sizes=sizes, pieces=pieces,
                    piece_batches=piece_batches, results=results, gathered=gathered, gathered_meta=gathered_meta, n_launches=n_launches,
                    report=report, gather_ms=gather_ms, gather_host_ms=gather_host_ms)

    def check_run(r):
        """Every pass over a pool batch must reproduce its first result bit for bit, and rank 0's own columns of the gathered logits
        (and rows of the gathered meta weights) must be its own results (checked outside the timed region).  Every rank puts the
        checksum of the columns IT produced into its report (`own_checksum`); rank 0 keeps the checksums of the columns it RECEIVED
        for every rank (`received_checksums`): after the reports are collected the two are compared rank by rank
        (`gather_verified_ranks`).  Drops the run's host results.  -> (drift, finite)."""
        res, got, got_meta = r.pop("results"), r.pop("gathered"), r.pop("gathered_meta")
        first, bad = {}, 0
        for (s_, p), logits, meta, post in res:
            if p not in first:
                first[p] = (logits, meta, post)
            elif not (np.array_equal(logits, first[p][0]) and np.array_equal(post, first[p][2])
                      and (meta is None or np.array_equal(meta, first[p][1]))):
                bad += 1
        mine = np.concatenate([lg for _, lg, _, _ in res], axis=1) if res else np.zeros((eng.n_experts, 0), np.float32)
        mine_meta = (np.concatenate([m for _, _, m, _ in res], axis=0) if res else np.zeros((0, 3), np.float32)) if eng.has_meta else None
        r["report"]["own_checksum"] = {"logits": checksum(mine), "meta": None if mine_meta is None else checksum(mine_meta)}
        if got is not None:
            bad += 0 if np.array_equal(got[:, :mine.shape[1]].numpy(), mine) else 1
            if got_meta is not None:
                bad += 0 if np.array_equal(got_meta[:mine_meta.shape[0]].numpy(), mine_meta) else 1
            received, col, row = [], 0, 0
            for s, a in r["sizes"]:
                s, a = s * args.steps, a * args.steps
                received.append({"logits": checksum(got[:, col:col + a].numpy()),
                                 "meta": None if got_meta is None else checksum(got_meta[row:row + s].numpy())})
                col, row = col + a, row + s
            r["received_checksums"] = received
        ok = all(np.isfinite(lg).all() and np.isfinite(po).all() for _, lg, _, po in res[:len(r["piece_batches"])])
        return bad, ok

    def verify_gather(r, reports):
        """-> number of ranks whose own checksum equals the checksum of what rank 0 received for them (None without a gather)."""
        received = r.get("received_checksums")
        if received is None:
            return None
        by_rank = {int(x.get("rank", 0)): x.get("own_checksum") for x in reports}
        return int(sum(1 for k, rc in enumerate(received) if by_rank.get(k) == rc))

    headline_mode = "strong" if args.scaling == "strong" else "weak"
    run = timed_region(headline_mode, profile=True)
    dt, value, sizes, pieces, piece_batches = run["dt"], run["value"], run["sizes"], run["pieces"], run["piece_batches"]
    n_launches, sites_total = run["n_launches"], run["sites_total"]
    pinned_bytes = run["report"]["pinned_input_bytes"]

    # ---- the dominant kernel's launch time over the timed region (HIP events on the launch stream) ---------
    op_rows, n_fw = eng.op_times_ms()
    eng.set_profiling(0)
    fused = eng.program.fused_read_convolver
    dom = readconv_roofline(eng, op_rows, piece_batches)
    dom_ms, dom_flops, dom_exec = dom["launch_ms"], dom["flops"], dom["exec_flops"]
    reads_per_launch = float(np.mean([n_reads(b) for b in piece_batches]))
    algorithmic = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    executed = dom_exec / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    drift, finite = check_run(run)

    # HBM bytes from the committed PMC passes of the same command (profiles/hbm_traffic.json, tools/profile_round.sh): the
    # dominant kernel per launch (`traffic`, the contract's field) and EVERY kernel of the forward (`forward_traffic`) --
    # reported only when that file was measured on THIS library (sha256 of the .so this process loaded) AND this configuration
    # (the passes profile the default C2 run), else null + stale
    traffic, forward_traffic, traffic_prov = committed_traffic(os.path.join(ROOT, "profiles", "hbm_traffic.json"), lib_sha256())
    if args.config != "C2":
        traffic, forward_traffic = None, None
        traffic_prov["traffic_stale"] = True
        traffic_prov["note"] = "the committed PMC passes profile the default configuration (C2)"
    flops_launch = float(np.mean([program_flops(eng.program, pool[k].site_slice(lo, hi))[0] for k, lo, hi in pieces]))
    launch_s = dt / n_launches
    per_read_macs = [o.macs_per_row for o in eng.program.ops if o.kind == 8]
    roofline = {
        "bound": "mfma", "kernel": dom["kernel"],
        # `achieved` = the FLOPs the matrix cores EXECUTE per launch / the kernel's average launch duration: the k3/s1
        # convolutions run in Winograd form (F(3,3): 5 instead of 9 fp32 contractions per 3 positions; F(2,3): 4
        # instead of 6 per 2), so the hardware fraction is priced on executed MFMA work and stays <= 1
        "achieved": round(executed, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "forward_traffic": forward_traffic,
        "traffic_stale": traffic_prov["traffic_stale"], "traffic_provenance": traffic_prov,
        # the same launch priced on ALGORITHMIC work (direct-form 2 * MAC, SURVEY.md 8d: 10.152 MFLOP per read)
        "algorithmic_achieved": round(algorithmic, 3), "algorithmic_frac": round(algorithmic / FP32_MFMA_PEAK_TFLOPS, 4),
        "formulas": {"frac": "2 * executed MAC per read * reads per launch / launch_ms / 157.3 TFLOP/s",
                     "algorithmic_frac": f"2 * {' | '.join(str(m) for m in per_read_macs) or 'direct-form'} MAC per read * reads per launch / launch_ms / "
                                         "157.3 TFLOP/s (exceeds 1 because Winograd executes fewer MFMA FLOPs than the direct form)"},
        # `launch` = one forward's run of the dominant kernel: since round 2 that is TWO launches of readconv_kernel per read
        # technology (whole rounds of 8-group workgroups, then one-group workgroups: readconv_plan) followed by its small finalize
        # kernel; launch_ms is the device time from the first launch's start to the finalize's end (two HIP events), summed over the
        # model's read technologies
        "launch_ms": round(dom_ms, 4), "launches_timed": int(n_fw), "kernel_launches_per_forward": 2 * dom["n_ops"] if dom["fused"] else 1,
        "reads_per_launch": round(reads_per_launch, 1),
        "flop_per_launch": float(dom_flops), "executed_flop_per_launch": float(dom_exec),
        "arithmetic": eng.program.arithmetic,          # the engine's own record, never a literal
        "form": "k3/s1 convolutions in Winograd form (residual trunk and allele stage F(3,3), stem F(2,3))" if eng.program.winograd else "direct form",
        "whole_launch_algorithmic_frac": round(flops_launch / launch_s / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
        "hbm_algorithmic_gbs": round(float(np.mean([batch_input_bytes(b) for b in piece_batches])) / launch_s / 1e9, 3),
        "pcie_h2d_gbs": round(float(np.mean([batch_input_bytes(b) for b in piece_batches])) / launch_s / 1e9, 3),
    }

    device_resident = latency = small = parity = two_engines = bf16x3 = configs = per_site_shared = None
    if secondary:
        stream = torch.cuda.current_stream(dev).cuda_stream
        # ---- device-resident rate: pileups already in HBM, outputs left there (no PCIe in the loop) --------------
        try:
            res = [dict(batch=b, dev=to_device(b, dev), out=device_outputs(eng, b, dev)) for b in pool[:2]]

            def dstep(i):
                r = res[i % len(res)]
                eng.forward_batch(r["dev"], stream=stream, out=r["out"], posteriors=True)
            for i in range(3):
                dstep(i)
            torch.cuda.synchronize(dev)
            n_dev = 40
            t1 = time.perf_counter()
            for i in range(n_dev):
                dstep(i)
            torch.cuda.synchronize(dev)
            dt_dev = time.perf_counter() - t1
            # per-op device times from a separate, untimed pass (an event record around every op)
            eng.set_profiling(10)
            for i in range(10):
                dstep(i)
            torch.cuda.synchronize(dev)
            rows, n_prof = eng.op_times_ms()
            eng.set_profiling(0)
            if args.op_times:
                for i, (k, n, ms) in enumerate(rows):
                    print(f"  op {i:3d} {k:15s} {ms:9.4f} ms  {n}", file=sys.stderr)
                print(f"  sum of ops {sum(r[2] for r in rows):.3f} ms over {n_prof} forwards", file=sys.stderr)
            # everything behind the read convolver, priced like the dominant kernel (executed / algorithmic MFMA work)
            r0 = res[0]["batch"]
            stage = [(op, ms) for op, (k, n, ms) in zip(eng.program.ops, rows) if op.kind != 8]
            stage_ms = sum(ms for _, ms in stage)
            stage_exec = sum(2.0 * (op.exec_macs_per_row or op.macs_per_row) * _rows_of(op, r0) for op, _ in stage)
            stage_alg = sum(2.0 * op.macs_per_row * _rows_of(op, r0) for op, _ in stage)
            roofline["allele_stage"] = {
                "ms_per_launch": round(stage_ms, 4), "kernel_launches": len(stage),
                "executed_tflops": round(stage_exec / (stage_ms * 1e-3) / 1e12, 2) if stage_ms > 0 else None,
                "executed_frac": round(stage_exec / (stage_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if stage_ms > 0 else None,
                "algorithmic_frac": round(stage_alg / (stage_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if stage_ms > 0 else None,
                "note": "every op after the fused read convolver (compressor, expert, head, glue), per 8 192-site launch, "
                        "from the per-op HIP events of a separate profiled pass"}
            kernels = {}
            for op, (k, n, ms) in zip(eng.program.ops, rows):
                kname = {"conv1d": "conv1d_mfma_kernel", "readconv_fused": "readconv_kernel"}.get(k, k + "_kernel")
                if k == "conv1d" and (op.flags & 32):
                    kname = "conv1d_wino_kernel"
                ent = kernels.setdefault(kname, dict(ms=0.0, launches=0))
                ent["ms"] += ms
                ent["launches"] += 1
            device_resident = {
                "value": round(args.sites * n_dev / dt_dev, 1), "unit": "sites/s", "sites_per_launch": args.sites,
                "launches": n_dev, "ms_per_launch": round(1e3 * dt_dev / n_dev, 4),
                "kernels_ms_per_launch": {kk: round(v["ms"], 4) for kk, v in sorted(kernels.items(), key=lambda x: -x[1]["ms"])},
                "kernel_launches": {kk: v["launches"] for kk, v in kernels.items()},
                "note": "inputs resident in HBM, outputs left on the device; per-op table from a separate profiled pass"}
            del res
        except Exception as exc:
            print(f"device-resident leg failed: {exc!r}", file=sys.stderr)

        # ---- the same host-to-host stream with TWO engines alternating (own scratch and compute stream each): the
        # second engine's kernels fill the tails of the first one's launches (9th round of the compressor's workgroups,
        # last round of the read convolver's).  Not the headline: concurrent kernels stretch each other's durations,
        # which would blur the per-kernel roofline above.
        try:
            pinned2 = [pinned(k, 0, pool[k].n_sites) for k in range(len(pool))]
            eng2 = Engine(spec, state, device=dev_index)
            pipe2 = HostPipeline(engines=[eng, eng2], posteriors=True)
            done = 0
            for i in range(6):
                done += len(pipe2.submit(pinned2[i % len(pinned2)], tag=i))
            done += len(pipe2.flush())
            torch.cuda.synchronize(dev)
            n_two = 60
            t1 = time.perf_counter()
            got = 0
            for i in range(n_two):
                got += len(pipe2.submit(pinned2[i % len(pinned2)], tag=i))
            got += len(pipe2.flush())
            torch.cuda.synchronize(dev)
            dt_two = time.perf_counter() - t1
            assert got == n_two
            two_engines = {"value": round(args.sites * n_two / dt_two, 1), "unit": "sites/s", "engines": 2, "launches": n_two,
                           "sites_per_launch": args.sites, "note": "HostPipeline(engines=[e1, e2]), host to host"}
            eng2.close()
        except Exception as exc:
            print(f"two-engine leg failed: {exc!r}", file=sys.stderr)

        # ---- the selectable arithmetic modes (read convolver's residual trunk on the bf16 matrix cores as 3-term splits,
        # fp32 residual stream): NOT the headline -- `value` is exact fp32 -- reported beside it with their own parity
        bf16x3 = {}
        for mode in (("bf16x3", "bf16x3+32") if args.config == "C2" else ()):
            try:
                engb = Engine(spec, state, device=dev_index, arithmetic=mode)
                resb = [dict(dev=to_device(b, dev), out=device_outputs(engb, b, dev)) for b in pool[:2]]

                def bstep(i):
                    r = resb[i % len(resb)]
                    engb.forward_batch(r["dev"], stream=stream, out=r["out"], posteriors=True)
                for i in range(3):
                    bstep(i)
                torch.cuda.synchronize(dev)
                engb.set_profiling(40, only="readconv_fused")
                t1 = time.perf_counter()
                for i in range(40):
                    bstep(i)
                torch.cuda.synchronize(dev)
                dt_b = time.perf_counter() - t1
                rows_b, _ = engb.op_times_ms()
                engb.set_profiling(0)
                kernel_ms = max(r[2] for r in rows_b)
                pipe_b = HostPipeline(engb, depth=2, posteriors=True)
                for i in range(4):
                    pipe_b.submit(piece_batches[i % len(piece_batches)], tag=i)
                pipe_b.flush()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                got = 0
                for i in range(60):
                    got += len(pipe_b.submit(piece_batches[i % len(piece_batches)], tag=i))
                got += len(pipe_b.flush())
                torch.cuda.synchronize(dev)
                dt_h = time.perf_counter() - t1
                split_layers = 7 if mode == "bf16x3" else 13
                entry = {"device_resident": round(args.sites * 40 / dt_b, 1), "host_to_host": round(args.sites * got / dt_h, 1),
                         "unit": "sites/s", "ms_per_launch": round(1e3 * dt_b / 40, 4), "readconv_launch_ms": round(kernel_ms, 4),
                         "split_convolutions": split_layers}
                if args.config in answers:
                    entry["parity"] = parity_of(engb, answers[args.config])
                bf16x3[mode] = entry
                del resb
                engb.close()
            except Exception as exc:
                print(f"{mode} leg failed: {exc!r}", file=sys.stderr)
        bf16x3["arithmetic"] = ("read convolver: stem, strided convolution + shortcut and per-allele sums exact fp32; the seven 64 -> 64 "
                                "residual-trunk convolutions ('bf16x3') and the six 32 -> 32 ones too ('bf16x3+32') as x w ~= xh wh + xh wl "
                                "+ xl wh on v_mfma_f32_16x16x32_bf16 with the residual stream kept in fp32 registers; allele stage exact fp32")
        bf16x3["note"] = "Engine(..., arithmetic=...): selectable, never the default; `value` above is exact fp32"

        # ---- latency: the reference's deployment form is ONE site per call (caller_calling.py:872-891) ------------
        try:
            from hello_amd.wrapper import ScoringNetwork
            net = ScoringNetwork(spec, state, device=dev_index, providePredictions=True)
            lb = make_config_sites(args.config, 256, args.seed + 77)
            site_args = [({a: (torch.from_numpy(f), None if g is None else torch.from_numpy(g)) for a, (f, g) in fd.items()},
                          torch.from_numpy(seg)) for fd, seg in feature_dicts(lb)]
            for fd, seg in site_args[:16]:
                net(fd, seg)
            t1 = time.perf_counter()
            for fd, seg in site_args:
                net(fd, seg)
            per_site = (time.perf_counter() - t1) / len(site_args)

            def launch_ms(n, on_device, reps):
                sub = lb.site_slice(0, n)
                sub = to_device(sub, dev) if on_device else sub
                for _ in range(3):
                    net.engine.forward_batch(sub, posteriors=True)
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                for _ in range(reps):
                    net.engine.forward_batch(sub, posteriors=True)
                torch.cuda.synchronize(dev)
                return 1e3 * (time.perf_counter() - t) / reps
            latency = {
                "per_site_call_ms": round(1e3 * per_site, 4), "per_site_calls_per_s": round(1.0 / per_site, 1),
                "per_site_form": "network(featureDict, ref_segment) -> 5-tuple, float tensors in, one engine, host in/out",
                "launch_1_site_host_ms": round(launch_ms(1, False, 50), 4),
                "launch_16_sites_host_ms": round(launch_ms(16, False, 50), 4),
                "launch_256_sites_host_ms": round(launch_ms(256, False, 30), 4),
                "launch_16_sites_device_ms": round(launch_ms(16, True, 50), 4),
                "launch_256_sites_device_ms": round(launch_ms(256, True, 50), 4),
                "note": "ONE engine, one stream; *_host: NumPy arrays in, NumPy logits + posteriors out (synchronous); "
                        "*_device: resident input, back-to-back asynchronous launches (steady-state ms per launch)"}
            latency["launch_256_sites_device_sites_per_s"] = round(256e3 / latency["launch_256_sites_device_ms"], 1)
            net.close()
        except Exception as exc:
            print(f"latency leg failed: {exc!r}", file=sys.stderr)

        # BASELINE.json's config 2 names launches of 256 sites: such a launch cannot fill the chip alone, so small
        # batches are alternated over four engines (own scratch and stream each); reported beside the headline value
        try:
            small_engines = [eng] + [Engine(spec, state, device=dev_index) for _ in range(3)]
            streams = [torch.cuda.Stream(dev) for _ in small_engines]
            sb = to_device(make_config_sites(args.config, 256, args.seed + 77), dev)
            torch.cuda.synchronize(dev)

            def small_step(i):
                k = i % len(small_engines)
                with torch.cuda.stream(streams[k]):
                    small_engines[k].forward_batch(sb, stream=streams[k].cuda_stream, posteriors=True)
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

        if args.config in answers:
            parity = parity_of(eng, answers[args.config])

        # ---- the reference's worker pool on this card (call.py:111,215-221): W single-threaded worker PROCESSES, each with the
        # unchanged per-site call, sharing ONE scoring server (loader.load(path, shared=True), hello_amd/shared.py +
        # csrc/site_server.hip).  A child process tree of its own (fresh interpreters; the workers never touch the GPU).
        if not args.no_shared_leg:
            try:
                import subprocess
                workers = max(2, min(16, host_cores()))
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "per_site_multiprocess.py"), "--shared", "--json", "--workers",
                                      str(workers), "--calls", "1500", "--config", args.config],
                                     cwd=ROOT, capture_output=True, text=True, timeout=240)
                lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                if out.returncode == 0 and lines:
                    per_site_shared = json.loads(lines[-1])
                    per_site_shared["note"] = ("worker processes of the reference's deployment form, one site per call each through the unchanged "
                                               "plug-in surface, scored by one shared server process on this GPU")
                else:
                    print(f"shared-server leg failed (rc {out.returncode}): {out.stderr[-800:]}", file=sys.stderr)
            except Exception as exc:
                print(f"shared-server leg failed: {exc!r}", file=sys.stderr)

        # ---- the OTHER BASELINE.json configurations, through the same path (make_sites -> pinned host batch -> HostPipeline ->
        # Engine, posteriors back on the host): host-to-host rate at --sites sites per launch, the dominant kernel's roofline
        # fraction from the engine's HIP events, max |delta| vs the oracle on the check sites
        if other_configs:
            configs = {}
            for name in other_configs:
                try:
                    t_leg = time.perf_counter()
                    c = BENCH_CONFIGS[name]
                    spec_c = ns.build(c["spec"])
                    eng_c = Engine(spec_c, weights.synth_state(spec_c, seed=args.seed), device=dev_index, arithmetic="fp32")
                    raw = make_config_sites(name, args.sites, 2000 + args.seed)
                    host = pin_batch(raw)
                    pipe_c = HostPipeline(eng_c, depth=2, posteriors=True)
                    for i in range(3):
                        pipe_c.submit(host, tag=i)
                    pipe_c.flush()
                    torch.cuda.synchronize(dev)
                    n_c = max(args.config_launches, 1)
                    eng_c.set_profiling(n_c, only="readconv_fused")
                    outs = []
                    t1 = time.perf_counter()
                    for i in range(n_c):
                        outs += pipe_c.submit(host, tag=i)
                    outs += pipe_c.flush()
                    torch.cuda.synchronize(dev)
                    dt_c = time.perf_counter() - t1
                    rows_c, n_fw_c = eng_c.op_times_ms()
                    eng_c.set_profiling(0)
                    assert len(outs) == n_c
                    same = all(np.array_equal(o[1], outs[0][1]) and np.array_equal(o[3], outs[0][3])
                               and (o[2] is None or np.array_equal(o[2], outs[0][2])) for o in outs[1:])
                    d = readconv_roofline(eng_c, rows_c, [raw])
                    alg_c, exe_c = program_flops(eng_c.program, raw)
                    ms_c = 1e3 * dt_c / n_c
                    configs[name] = {
                        "workload": c["label"], "model": c["spec"], "value": round(args.sites * n_c / dt_c, 1), "unit": "sites/s",
                        "sites_per_launch": args.sites, "launches": n_c, "ms_per_launch": round(ms_c, 4),
                        "reads_per_site": round((int(raw.reads0.shape[0]) + (0 if raw.reads1 is None else int(raw.reads1.shape[0]))) / raw.n_sites, 2),
                        "alleles_per_site": round(raw.n_alleles / raw.n_sites, 3), "channels": [eng_c.program.channels0, eng_c.program.channels1],
                        "n_experts": eng_c.n_experts, "has_meta": bool(eng_c.has_meta),
                        "roofline_frac": round(d["exec_flops"] / (d["launch_ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        "roofline_algorithmic_frac": round(d["flops"] / (d["launch_ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        "roofline": {"bound": "mfma", "kernel": d["kernel"], "launch_ms": round(d["launch_ms"], 4), "launches_timed": int(n_fw_c),
                                     "read_technologies": d["n_ops"], "reads_per_launch": d["rows"], "flop_per_launch": d["flops"],
                                     "executed_flop_per_launch": d["exec_flops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"},
                        "whole_launch_algorithmic_frac": round(alg_c / (ms_c * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        "whole_launch_executed_frac": round(exe_c / (ms_c * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        "mflop_per_site": round(alg_c / raw.n_sites / 1e6, 1),
                        "repeat_passes_bit_identical": bool(same), "arithmetic": eng_c.program.arithmetic,
                        "parity": parity_of(eng_c, answers[name]) if name in answers else None,
                        "leg_seconds": None}
                    eng_c.close()
                    del host, raw, pipe_c, outs
                    torch.cuda.empty_cache()
                    configs[name]["leg_seconds"] = round(time.perf_counter() - t_leg, 2)
                except Exception as exc:
                    print(f"configuration leg {name} failed: {exc!r}", file=sys.stderr)
                    configs[name] = {"value": None, "error": repr(exc)}
            configs["note"] = ("the other BASELINE.json configurations through the headline's path on this GPU: host-resident pileups -> "
                               "host-resident logits + posteriors (HostPipeline), one pinned batch of --sites sites cycled; "
                               "`python bench.py --config <name> [--gpus N]` runs any of them as the timed region itself")

    # ---- who ran where (after the timed region): every rank's own report, one all_gather_object -----------------------
    raw_reports = shard.collect_rank_reports(run["report"])
    gather_verified = verify_gather(run, raw_reports)
    reports = shard.summarize_ranks(raw_reports)
    strong = None
    if args.scaling == "both":
        if world > 1:
            # a second timed region of the same K steps: the N = 1 stream cut N ways (total work fixed)
            run2 = timed_region("strong", profile=False)
            drift2, finite2 = check_run(run2)
            drift += drift2
            finite = finite and finite2
            raw2 = shard.collect_rank_reports(run2["report"])
            rep2 = shard.summarize_ranks(raw2)
            strong = {"value": round(run2["value"], 1), "unit": "sites/s", "scaling": "strong",
                      "ms_per_step": round(1e3 * run2["dt"] / max(args.steps, 1), 4), "sites_total": int(run2["sites_total"]),
                      "timed_region_s": round(run2["dt"], 3), "gather_ms": run2["gather_ms"], "gather_host_ms": run2["gather_host_ms"],
                      "slowest_rank": rep2["slowest_rank"], "balance": rep2["balance"], "distinct_devices": rep2["distinct_devices"],
                      "repeat_passes_bit_identical": drift2 == 0, "outputs_finite": bool(finite2),
                      "gather_verified_ranks": verify_gather(run2, raw2), "ranks": rep2["ranks"]}
            del run2
        else:
            strong = {"value": round(value, 1), "unit": "sites/s", "scaling": "strong",
                      "note": "N = 1: the strong and the weak workload are the same stream; one timed region"}
    cpu_reason = None
    if cpu is None:
        cpu_reason = ("skipped: --no-cpu-baseline" if args.no_cpu_baseline else
                      "skipped: running under a profiler whose preloaded library has initialised the GPU (the baseline forks)" if profiled else
                      "not run on this rank")

    if rank == 0:
        b0 = pool[0]
        scaling_label = headline_mode
        print(f"bench: N = {world} ({backend if dist is not None else 'no process group'}), config {args.config}, `value` = {scaling_label.upper()} scaling"
              + (f", strong scaling beside it: {strong['value']:.0f} sites/s" if strong and world > 1 else "")
              + f"; {reports['distinct_devices']} distinct device(s) over {reports['ranks_seen']} rank(s)"
              + (f"; gather verified for {gather_verified} of {world} rank(s)" if gather_verified is not None else ""), file=sys.stderr)
        line = {
            "metric": "candidate sites/sec (whole node)", "value": round(value, 1), "unit": "sites/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / max(args.steps, 1), 4), "higher_is_better": True, "scaling": scaling_label,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg['label']}, seeded synthetic weights; host-resident uint8 pileups + counts -> "
                                   f"host-resident logits + genotype-pair posteriors (PCIe both ways inside the timed "
                                   f"region) through shard.partition_sites + HostPipeline; a step = "
                                   f"{args.launches_per_step} launches x {args.sites} sites per GPU, cycling a pinned "
                                   f"pool of {args.pool} distinct batches",
                       "name": args.config, "model": cfg["spec"],
                       "sites_per_launch": args.sites, "launches_per_step_per_gpu": args.launches_per_step,
                       "sites_per_step": int(sum(s for s, _ in sizes)), "sites_total": int(sites_total),
                       "timed_region_s": round(dt, 3),
                       "reads_per_site": round(n_reads(b0) / b0.n_sites, 2),
                       "alleles_per_site": round(b0.n_alleles / b0.n_sites, 3),
                       "window": int(eng.program.window), "channels": int(eng.program.channels0),
                       "channels_second_technology": int(eng.program.channels1) or None,
                       "n_experts": int(eng.n_experts), "has_meta": bool(eng.has_meta),
                       "parallelism": f"site-sharded dp{world}, one gather at the end",
                       "arithmetic": eng.program.arithmetic,
                       "fused_read_convolver": bool(fused), "outputs": "logits + genotype-pair posteriors (host)",
                       "host_cpus_of_rank0": len(cpus), "pinned_input_bytes_of_rank0": pinned_bytes,
                       "repeat_passes_bit_identical": drift == 0, "outputs_finite": bool(finite)},
            "roofline": roofline,
            "cpu_baseline": cpu, "cpu_baseline_reason": cpu_reason,
            # audit of the run: one entry per rank (device PCI address / UUID, sites, reads, own seconds, launches, pinned bytes,
            # CPUs, checksum of the columns it produced), collected with one all_gather_object after the closing fence
            "backend": (backend if dist is not None else None), "ranks_seen": reports["ranks_seen"],
            "distinct_devices": reports["distinct_devices"], "identity_sources": reports["identity_sources"],
            "identity_warning": reports["identity_warning"], "slowest_rank": reports["slowest_rank"],
            "rank_seconds_min_max": reports["rank_seconds_min_max"], "balance": reports["balance"],
            "gather_ms": run["gather_ms"], "gather_host_ms": run["gather_host_ms"],
            # the gather proves itself: for how many ranks the checksum of what rank 0 RECEIVED equals the checksum that rank
            # computed locally of what it PRODUCED (crc32 + float64 sums + first / last value; null when no process group is up)
            "gather_verified_ranks": gather_verified,
            "ranks": reports["ranks"],
            "strong_scaling": strong,
            "device_resident": device_resident,
            "two_engines": two_engines,
            "bf16x3": bf16x3,
            "latency": latency,
            "parity": parity,
            "small_batch": small,
            "per_site_shared": per_site_shared,
            "configs": configs,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if drift:
        raise SystemExit(f"{drift} repeated passes differed from the first pass over the same batch")
    if gather_verified is not None and rank == 0 and gather_verified != world:
        raise SystemExit(f"the gather delivered the columns of only {gather_verified} of {world} ranks intact")


if __name__ == "__main__":
    main()
