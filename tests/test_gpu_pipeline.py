"""The host->device streaming pipeline returns exactly what a direct engine call returns, in order."""
import numpy as np
import pytest

from hello_amd import netspec as ns, synth, weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,kw", [
    ("single_tech", dict(coverage=30)),
    ("hybrid_ensemble2", dict(coverage=20, hybrid_coverage=10)),      # two read sets, one-hot reference, meta
])
def test_pipeline_matches_direct_calls_in_order(cfg, kw):
    import torch
    from hello_amd.engine import Engine
    from hello_amd.pipeline import HostPipeline
    spec = ns.build(cfg)
    state = weights.synth_state(spec, seed=5)
    eng = Engine(spec, state, device=0)
    batches = [synth.make_sites(n, seed=40 + i, **kw) for i, n in enumerate([64, 7, 200, 33, 1, 120])]
    direct = [eng.forward_batch(b, posteriors=True) for b in batches]
    pipe = HostPipeline(eng, depth=2)
    got = []
    for i, b in enumerate(batches):
        if i == 2:            # a producer that already writes pinned memory: no staging copy
            b = synth.SiteBatch(torch.from_numpy(b.reads0).pin_memory(), b.reads_per_allele0, b.alleles_per_site,
                                b.ref_onehot, b.reads1, b.reads_per_allele1)
        got += pipe.submit(b, tag=i)
    got += pipe.flush()
    assert [g[0] for g in got] == list(range(len(batches)))
    for (tag, logits, meta, post), (dl, dm, dp) in zip(got, direct):
        assert np.array_equal(logits, dl) and np.array_equal(post, dp)
        assert (meta is None and dm is None) or np.array_equal(meta, dm)
    assert pipe.flush() == []
    # several engines: consecutive batches run concurrently on their own compute streams, same answers, same order
    eng2 = Engine(spec, state, device=0)
    pipe = HostPipeline(engines=[eng, eng2])
    got = []
    for i, b in enumerate(batches * 2):
        got += pipe.submit(b, tag=i)
    got += pipe.flush()
    assert [g[0] for g in got] == list(range(2 * len(batches)))
    for (tag, logits, meta, post), (dl, dm, dp) in zip(got, direct * 2):
        assert np.array_equal(logits, dl) and np.array_equal(post, dp)
    eng2.close()
    eng.close()


def test_pipeline_sink_keeps_logits_resident():
    """The multi-GPU path's device sink: every batch's logits (and meta) also land in a caller-owned device tensor
    at the given allele column / site row, bit-identical to what the pipeline returns to the host."""
    import torch
    from hello_amd.engine import Engine
    from hello_amd.pipeline import HostPipeline, pin_batch
    spec = ns.build("hybrid_ensemble2")
    state = weights.synth_state(spec, seed=5)
    eng = Engine(spec, state, device=0)
    whole = pin_batch(synth.make_sites(90, seed=61, coverage=20, hybrid_coverage=10))
    cuts = [0, 40, 41, 90]
    pieces = [whole.site_slice(a, b) for a, b in zip(cuts, cuts[1:])]          # pinned views, no staging copy
    assert all(p.reads0.is_pinned() and p.reads0.is_contiguous() for p in pieces)
    sink_l = torch.zeros((eng.n_experts, whole.n_alleles), device="cuda")
    sink_m = torch.zeros((whole.n_sites, 3), device="cuda")
    pipe = HostPipeline(eng, depth=2)
    got, col, row = [], 0, 0
    for p in pieces:
        got += pipe.submit(p, sink=(sink_l, sink_m, col, row))
        col, row = col + p.n_alleles, row + p.n_sites
    got += pipe.flush()
    torch.cuda.synchronize()
    assert np.array_equal(sink_l.cpu().numpy(), np.concatenate([g[1] for g in got], axis=1))
    assert np.array_equal(sink_m.cpu().numpy(), np.concatenate([g[2] for g in got], axis=0))
    direct = eng.forward_batch(synth.SiteBatch(whole.reads0.numpy(), whole.reads_per_allele0, whole.alleles_per_site,
                                               whole.ref_onehot.numpy(), whole.reads1.numpy(), whole.reads_per_allele1))
    np.testing.assert_allclose(sink_l.cpu().numpy(), direct[0], rtol=1e-5, atol=1e-5)   # per-site sums re-associate
    eng.close()


def _bench_env():
    import os
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HELLO_BENCH_BACKEND"):
        env.pop(k, None)
    return env


def _json_lines(text):
    return [ln for ln in text.splitlines() if ln.startswith("{")]


SMALL = ["--steps", "3", "--warmup", "1", "--sites", "512", "--launches-per-step", "2", "--no-secondary"]


def test_bench_under_torchrun_is_auditable():
    """bench.py as the driver launches it for N > 1 (python -m torch.distributed.run, one process per GPU), here with one rank on
    this box's one GPU and a small workload: a fresh child process (the launcher runs before anything touches the GPU), RCCL
    communicator up, the product's partitioner + pipeline + the one gather on the measured path, ONE JSON line -- which carries the
    run's audit trail: the ranks seen with their device's PCI address / UUID and the identity source, sites, reads, own seconds,
    launches, pinned bytes and CPUs; distinct_devices == 1; the event-timed gather; the strong-scaling entry (--scaling both, the
    default); the sha256 of the library it loaded.  The same workload run plainly scores the same number of sites."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _bench_env()
    common = ["--gpus", "1"] + SMALL + ["--no-cpu-baseline"]
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    under = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                            "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), os.path.join(root, "bench.py")] + common,
                           cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert under.returncode == 0, under.stderr[-2000:]
    lines = _json_lines(under.stdout)
    assert len(lines) == 1, under.stdout[-2000:]
    a, b = json.loads(_json_lines(plain.stdout)[-1]), json.loads(lines[0])
    assert a["scaling"] == "weak" and b["scaling"] == "weak" and b["n_gpus"] == 1 and b["steps"] == 3
    assert a["config"]["sites_total"] == b["config"]["sites_total"] == 3 * 2 * 512 and b["config"]["repeat_passes_bit_identical"]
    assert b["value"] > 0 and 0 < b["roofline"]["frac"] <= 1.0 and b["roofline"]["kernel"] == "readconv_kernel"
    # the headline's arithmetic comes from the engine's own record
    assert a["dtype"] == b["dtype"] == "f32" and a["roofline"]["arithmetic"] == b["config"]["arithmetic"] == "fp32"
    # audit trail
    assert b["backend"] == "nccl" and b["ranks_seen"] == 1 and b["distinct_devices"] == 1 and b["slowest_rank"] == 0 and b["balance"] == 1.0
    (r,) = b["ranks"]
    assert r["rank"] == 0 and r["device_index"] == 0 and r["pci_bus_id"] and r["sites"] == 3 * 2 * 512 and r["launches"] == 6
    assert "pci_bus_id" in r["identity_source"] and b["identity_warning"] is None
    assert r["reads"] > 25 * r["sites"] and r["pinned_input_bytes"] > 0 and r["pinned_input_bytes"] % 900 == 0 and r["cpus_pinned"] >= 1
    assert 0 < r["timed_seconds"] <= b["config"]["timed_region_s"] + 1e-3 and b["gather_ms"] is not None and b["gather_ms"] >= 0
    assert b["strong_scaling"]["value"] == b["value"] and "N = 1" in b["strong_scaling"]["note"]
    assert b["gather_verified_ranks"] == 1 and a["gather_verified_ranks"] is None and b["config"]["name"] == "C2"     # RCCL at world 1: the gather checks itself
    assert a["backend"] is None and a["ranks_seen"] == 1 and a["distinct_devices"] == 1 and a["gather_ms"] is None
    # the CPU baseline was switched off here: null with its reason
    assert a["cpu_baseline"] is None and "--no-cpu-baseline" in a["cpu_baseline_reason"] and b["cpu_baseline"] is None
    # roofline.traffic is tied to the binary: the hash of the library this run loaded is on the line, and bytes appear only when
    # profiles/hbm_traffic.json was measured on it
    prov = b["roofline"]["traffic_provenance"]
    assert len(prov["lib_sha256"]) == 64 and (b["roofline"]["traffic"] is None) == bool(b["roofline"]["traffic_stale"] in (True, None))


def test_bench_gpus_2_from_a_plain_shell_starts_its_own_ranks():
    """`python bench.py --gpus 2` typed into a plain shell (no launcher, no RANK in the environment) starts its two
    ranks itself as a child torch.distributed.run, and prints rank 0's ONE JSON line -- here with HELLO_BENCH_BACKEND=gloo so that
    both ranks can share this box's one GPU.  The line carries what the contract asks of every line, at N > 1 too: `roofline` and
    `cpu_baseline` (timed on rank 0 before the rendezvous), plus the audit of both ranks and the strong-scaling second region
    (exercised here at world 2, with its own repeat / gather checks)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_bench_env(), HELLO_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-budget", "2"] + SMALL, cwd=root, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1, out.stdout[-2000:]
    z = json.loads(lines[0])
    assert z["n_gpus"] == 2 and z["backend"] == "gloo" and z["ranks_seen"] == 2 and [r["rank"] for r in z["ranks"]] == [0, 1]
    assert z["distinct_devices"] == 1                                        # the rehearsal's two ranks share the one card, and say so
    assert z["scaling"] == "weak" and z["config"]["sites_total"] == 2 * 3 * 2 * 512 and z["config"]["repeat_passes_bit_identical"]
    assert sum(r["sites"] for r in z["ranks"]) == z["config"]["sites_total"] and 0.9 < z["balance"] <= 1.0
    assert z["roofline"]["kernel"] == "readconv_kernel" and 0 < z["roofline"]["frac"] <= 1.0 and z["roofline"]["launch_ms"] > 0
    cpu = z["cpu_baseline"]
    assert cpu is not None and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] == "port" and "rendezvous" in cpu["when"]
    assert z["cpu_baseline_reason"] is None
    st = z["strong_scaling"]
    assert st["scaling"] == "strong" and st["sites_total"] == 3 * 2 * 512 and [r["rank"] for r in st["ranks"]] == [0, 1]
    assert st["repeat_passes_bit_identical"] and st["outputs_finite"] and st["value"] > 0
    assert sum(r["sites"] for r in st["ranks"]) == st["sites_total"]
    assert z["gather_verified_ranks"] == 2 and st["gather_verified_ranks"] == 2
    assert "starting 2 ranks as a child" in out.stderr


@pytest.mark.parametrize("config,model,second,experts,meta", [("C4", "hybrid_no_ensemble", 6, 1, False),
                                                              ("hybrid_full", "hybrid_full", 6, 3, True)])
def test_bench_gpus_2_runs_the_hybrid_configurations_through_the_gather(config, model, second, experts, meta):
    """VERDICT r05 item 1 + 7: `python bench.py --gpus 2 --config C4` typed plainly (HELLO_BENCH_BACKEND=gloo: both ranks on this box's
    one GPU) runs BASELINE.json's hybrid configuration as the timed region -- two read technologies per rank, the one gather -- and the
    gather proves itself: every rank ships the checksum of the columns it produced, rank 0 compares it with the checksum of what it
    received (`gather_verified_ranks` == 2, in the weak and in the strong region).  With hybrid_full the three experts' logits AND the
    per-site meta weights cross the gather."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_bench_env(), HELLO_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", config, "--no-cpu-baseline"] + SMALL,
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1, out.stdout[-2000:]
    z = json.loads(lines[0])
    c = z["config"]
    assert z["n_gpus"] == 2 and z["backend"] == "gloo" and z["ranks_seen"] == 2 and c["name"] == config and c["model"] == model
    assert c["channels"] == 6 and c["channels_second_technology"] == second and c["n_experts"] == experts and c["has_meta"] is meta
    assert c["sites_total"] == 2 * 3 * 2 * 512 and c["repeat_passes_bit_identical"] and c["outputs_finite"] and c["reads_per_site"] > 40
    assert z["gather_verified_ranks"] == 2 and z["strong_scaling"]["gather_verified_ranks"] == 2
    for r in z["ranks"]:
        own = r["own_checksum"]
        assert own["logits"]["n"] == experts * r["alleles"] and (own["meta"] is not None) is meta
        if meta:
            assert own["meta"]["n"] == 3 * r["sites"] and abs(own["meta"]["sum"] - r["sites"]) < 1e-2 * r["sites"]      # softmax rows sum to 1
        assert r["pinned_input_bytes"] % 900 == 0 and r["reads"] > 40 * r["sites"]
    rf = z["roofline"]
    assert rf["kernel"] == "readconv_kernel" and rf["kernel_launches_per_forward"] == 4 and 0 < rf["frac"] <= 1.0
    assert rf["traffic"] is None and rf["traffic_stale"] is True               # the committed PMC passes profile C2
    assert z["configs"] is None and z["parity"] is None                        # secondary legs run at N = 1 only


def test_bench_default_run_reports_every_baseline_configuration():
    """the default (C2) run carries a `configs` block -- C3, C4, C5 and hybrid_full through the headline's path
    (pinned host batch -> HostPipeline -> Engine -> posteriors on the host): rate, the read convolver's roofline fraction from the
    engine's HIP events, and max |delta| against the oracle's per-site answers on the check sites.  Small sizes here; the driver's run
    uses 8 192 sites per launch over 40 launches."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--sites", "512", "--launches-per-step", "2",
                          "--cpu-budget", "1", "--config-launches", "4"], cwd=root, env=_bench_env(), capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    z = json.loads(_json_lines(out.stdout)[-1])
    assert z["config"]["name"] == "C2" and z["parity"]["within_tolerance"] and z["parity"]["sites"] == 96 and z["parity"]["sites_with_identical_call"] == 96
    ps = z["per_site_shared"]
    assert ps is not None and ps["value"] > 0 and ps["workers"] >= 2 and ps["server"]["server"] == "native" and ps["server"]["errors"] == 0
    assert ps["server"]["sites"] >= ps["workers"] * ps["calls_per_worker"] and ps["server"]["worker_gpu_fds"] == []
    assert z["cpu_baseline"]["config"] == "C2" and z["cpu_baseline"]["value"] > 0
    assert z["gather_verified_ranks"] is None                                  # no process group in a plain N = 1 run
    cfgs = z["configs"]
    assert {"C3", "C4", "C5", "hybrid_full"} <= set(cfgs)
    for name, model, techs in (("C3", "single_tech", 1), ("C4", "hybrid_no_ensemble", 2), ("C5", "single_tech_hp", 1), ("hybrid_full", "hybrid_full", 2)):
        e = cfgs[name]
        assert e["model"] == model and e["value"] > 0 and e["launches"] == 4 and e["sites_per_launch"] == 512, (name, e)
        assert 0 < e["roofline_frac"] <= 1.0 and e["roofline_algorithmic_frac"] > e["roofline_frac"] and e["roofline"]["read_technologies"] == techs
        assert e["repeat_passes_bit_identical"] and e["arithmetic"] == "fp32"
        par = e["parity"]
        assert par["within_tolerance"] and par["sites"] == 96 and par["max_abs_delta_pair_posterior"] <= 1e-4 and par["max_abs_delta_allele_probability"] <= 1e-4
        assert par["sites_with_identical_call"] == 96 and par["qual_max_abs_delta"] < 0.5
    assert cfgs["hybrid_full"]["parity"]["max_abs_delta_meta"] <= 1e-4 and cfgs["hybrid_full"]["n_experts"] == 3 and cfgs["hybrid_full"]["has_meta"]
    assert cfgs["C5"]["channels"] == [7, 0] and cfgs["C4"]["channels"] == [6, 6]


@pytest.mark.bench
def test_bench_under_torchrun_equals_the_plain_run():
    """The N = 1 point of a scaling curve agrees with the headline bench: the default workload under the launcher, its value inside
    the band of two plain runs taken before and after it on the same box (+- 10 %: a timing comparison on a shared card, kept out of
    the functional assertions above -- ADVICE r04)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _bench_env()
    common = ["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-secondary", "--no-cpu-baseline"]

    def run(prefix):
        out = subprocess.run(prefix + [os.path.join(root, "bench.py")] + common, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(_json_lines(out.stdout)[-1])
    a = run([sys.executable])
    b = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
             "--master-port", str(29300 + os.getpid() % 300)])
    a2 = run([sys.executable])
    assert a["config"]["sites_total"] == b["config"]["sites_total"] == a2["config"]["sites_total"] == 6 * 10 * 8192
    lo, hi = min(a["value"], a2["value"]), max(a["value"], a2["value"])
    assert 0.90 * lo <= b["value"] <= 1.10 * hi, (a["value"], b["value"], a2["value"])
