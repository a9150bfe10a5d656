"""Multi-GPU path on CPU: world_size-2 gloo processes shard a batch by sites, score their shard and
gather to rank 0 with the one collective of the path (hello_amd/shard.py)."""
import os
import sys

import numpy as np
import pytest

from hello_amd import netspec as ns, shard, synth, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_contiguous_complete_and_balanced():
    rng = np.random.default_rng(0)
    reads = rng.poisson(30, size=1000) + 1
    for parts in (1, 2, 3, 8):
        ranges = shard.partition_sites(reads, parts)
        assert ranges[0][0] == 0 and ranges[-1][1] == 1000
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        loads = [reads[lo:hi].sum() for lo, hi in ranges]
        assert max(loads) - min(loads) <= 2 * reads.max()
    assert shard.partition_sites([5, 5], 4)[-1][1] == 2           # fewer sites than ranks: empty ranges allowed
    assert shard.partition_sites([1000, 1, 1, 1], 2)[0] == (0, 1)  # ragged


def _worker(rank, world, port, out_path, use_oracle):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if use_oracle:
        from oracle import moe_oracle as mo
        spec = ns.build("hybrid_ensemble2")
        oracle = mo.Oracle(spec, weights.synth_state(spec, seed=2))
        batch = synth.make_sites(5, seed=8, coverage=6, hybrid_coverage=4)
        fn = lambda sub: mo.forward_batch(oracle, sub, chunk_sites=1)        # noqa: E731
    else:
        batch = synth.make_sites(3 if world > 3 else 37, seed=5, coverage=20)
        def fn(sub):     # a stand-in scorer: any per-allele function of the reads
            off = np.concatenate([[0], np.cumsum(sub.reads_per_allele0)])
            v = np.array([sub.reads0[off[i]:off[i + 1]].astype(np.float64).sum() for i in range(sub.n_alleles)])
            return v[None, :].astype(np.float32), None
    n_experts, has_meta = (3, True) if use_oracle else (1, False)
    calls = []
    real_gather, real_all_gather, real_all_reduce = dist.gather, dist.all_gather, dist.all_reduce
    dist.gather = lambda *a, **k: (calls.append("gather"), real_gather(*a, **k))[1]
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), real_all_gather(*a, **k))[1]
    dist.all_reduce = lambda *a, **k: (calls.append("all_reduce"), real_all_reduce(*a, **k))[1]
    logits, meta = shard.score_sharded(fn, batch, rank, world, n_experts=n_experts, has_meta=has_meta)
    dist.gather, dist.all_gather, dist.all_reduce = real_gather, real_all_gather, real_all_reduce
    assert calls == ["gather"], calls           # exactly one collective, whatever the model and the range
    if rank == 0:
        full_logits, full_meta = fn(batch)
        np.savez(out_path, got=logits.numpy(), want=np.asarray(full_logits),
                 got_meta=np.zeros(0) if meta is None else meta.numpy(),
                 want_meta=np.zeros(0) if full_meta is None else np.asarray(full_meta))
    else:
        assert logits is None and meta is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,use_oracle", [(2, False), (2, True), (4, False), (8, True), (8, False)])
def test_sharded_scoring_equals_unsharded(tmp_path, world, use_oracle):
    """World 8 on 5 (3) sites leaves ranks with empty ranges; the ensemble model adds meta rows to the same
    gather.  Sites are scored independently, so the sharded result equals the unsharded one bit for bit."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "out.npz")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, out, use_oracle), nprocs=world, join=True)
    z = np.load(out)
    assert np.array_equal(z["got"], z["want"])
    assert np.array_equal(z["got_meta"], z["want_meta"])


def test_shard_sizes_and_rank_cpus():
    aps = np.array([2, 1, 3, 2], np.int32)
    assert shard.shard_sizes(aps, [(0, 1), (1, 1), (1, 4)]) == [(1, 2), (0, 0), (3, 6)]
    allowed = sorted(os.sched_getaffinity(0))
    seen = []
    for r in range(4):
        cpus = shard.rank_cpus(r, 4)
        assert cpus and set(cpus) <= set(allowed)
        seen += cpus
    if len(allowed) >= 4:
        assert len(set(seen)) == len(seen)          # ranks do not share CPUs when there are enough
    # 8 ranks, GPUs 0-3 on NUMA node 0 and 4-7 on node 1: the four ranks of a node split ALL of that node's CPUs
    half = len(allowed) // 2
    topo = dict(node_of_device=lambda d: d // 4, cpus_of_node=lambda n: allowed[:half] if n == 0 else allowed[half:], n_devices=8)
    per_rank = [shard.rank_cpus(r, 8, r, **topo) for r in range(8)]
    if half >= 4:
        assert sorted(c for cpus in per_rank[:4] for c in cpus) == allowed[:half]
        assert sorted(c for cpus in per_rank[4:] for c in cpus) == allowed[half:]
    # two ranks rehearsing on ONE card share its node and split it
    one = dict(node_of_device=lambda d: 0, cpus_of_node=lambda n: allowed, n_devices=1)
    a, b = shard.rank_cpus(0, 2, 0, **one), shard.rank_cpus(1, 2, 0, **one)
    assert sorted(a + b) == allowed and (len(allowed) < 2 or not set(a) & set(b))


def test_bench_rank_pieces_cover_the_global_stream_exactly():
    """bench.py's multi-rank plan: the global step batch (launches cycling the pool) cut by partition_sites into one
    contiguous range per rank, handed out as (pool batch, site lo, site hi) pieces: every site of the stream exactly
    once, in order, and the sizes every rank computes locally add up."""
    sys.path.insert(0, ROOT)
    import bench
    pool = [synth.make_sites(n, seed=70 + i, coverage=12) for i, n in enumerate((90, 75, 110))]
    counts = [dict(reads_per_site=shard.reads_per_site(b), alleles_per_site=b.alleles_per_site) for b in pool]
    for world in (1, 2, 3, 8):
        launches = 4 * world
        seq = [i % 3 for i in range(launches)]
        total_sites = sum(pool[k].n_sites for k in seq)
        covered, sizes0 = [], None
        for rank in range(world):
            pieces, sizes = bench.rank_pieces(counts, launches, rank, world)
            sizes0 = sizes0 or sizes
            assert sizes == sizes0                                   # the same plan on every rank
            assert sum(hi - lo for _, lo, hi in pieces) == sizes[rank][0]
            assert sum(int(pool[k].alleles_per_site[lo:hi].sum()) for k, lo, hi in pieces) == sizes[rank][1]
            covered += pieces
        # concatenating the ranks' pieces walks the launch sequence front to back without gaps or overlaps
        pos, it = 0, iter(seq)
        k_cur, at = next(it), 0
        for k, lo, hi in covered:
            if at == pool[k_cur].n_sites:
                k_cur, at = next(it), 0
            assert k == k_cur and lo == at
            at = hi
            pos += hi - lo
        assert pos == total_sites == sum(s for s, _ in sizes0)
        reads = [sum(int(counts[k]["reads_per_site"][lo:hi].sum()) for k, lo, hi in bench.rank_pieces(counts, launches, r, world)[0])
                 for r in range(world)]
        assert max(reads) - min(reads) <= 2 * max(int(c["reads_per_site"].max()) for c in counts)     # balanced by reads


def _report_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # what bench.py's timed_region hands over: this rank's identity and clocks (two ranks rehearsing on ONE card share its address)
    mine = dict(rank=rank, local_rank=rank, host="box", pid=os.getpid(), device_index=0, pci_bus_id="0000:05:00.0", uuid="GPU-abc",
                name="MI355X", sites=100 + rank, reads=3000 + 30 * rank, alleles=210, launches=4, timed_seconds=1.0 + 0.25 * rank,
                pinned_input_bytes=900 * (3000 + 30 * rank), cpus_pinned=4, cpu_list="0-3")
    got = shard.summarize_ranks(shard.collect_rank_reports(mine))
    if rank == 0:
        json.dump(got, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_reports_make_a_multi_rank_bench_line_auditable(tmp_path):
    """after the timed region every rank's report travels with one all_gather_object; the summary names the
    ranks seen, the DISTINCT devices they drove, the slowest rank and the balance of the read-balanced partition."""
    import json
    import torch.multiprocessing as mp
    out = str(tmp_path / "reports.json")
    mp.spawn(_report_worker, args=(2, 29700 + (os.getpid() % 2000), out), nprocs=2, join=True)
    z = json.load(open(out))
    assert z["ranks_seen"] == 2 and [r["rank"] for r in z["ranks"]] == [0, 1]
    assert z["distinct_devices"] == 1                       # both ranks reported the same PCI address / UUID
    assert z["slowest_rank"] == 1 and z["rank_seconds_min_max"] == [1.0, 1.25]
    assert z["balance"] == round(3000 / 3030, 4)
    for r in z["ranks"]:
        assert {"rank", "device_index", "pci_bus_id", "uuid", "sites", "reads", "timed_seconds", "launches", "pinned_input_bytes",
                "cpus_pinned"} <= set(r)
    # eight ranks on eight cards of one host, one of them idle (an empty range): distinct by address, balance over the busy ones
    eight = [dict(rank=r, host="node", device_index=r, pci_bus_id=f"0000:{r:02x}:00.0", uuid=None, reads=0 if r == 7 else 1000 + r,
                  timed_seconds=2.0 - 0.1 * r) for r in range(8)]
    s = shard.summarize_ranks(reversed(eight))
    assert s["distinct_devices"] == 8 and s["slowest_rank"] == 0 and s["balance"] == round(1000 / 1006, 4)
    # without a process group: the one report, as a list
    assert shard.collect_rank_reports(dict(rank=0, reads=5)) == [dict(rank=0, reads=5)]
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_cpu_ranges([0, 1, 2, 5, 7, 8]) == "0-2,5,7-8" and bench.shard_cpu_ranges([]) == ""


def test_distinct_devices_is_not_counted_from_device_indices():
    """a rank that can name its card only by its index (no PCI address, no usable UUID) makes the count meaningless --
    ranks isolated by HIP_VISIBLE_DEVICES all see index 0, two ranks on one card see different ones.  The summary then reports
    None with a warning and says which identity source every rank used; PCI address and UUID are keyed TOGETHER."""
    named = [dict(rank=r, host="n", device_index=0, pci_bus_id=f"0000:{r:02x}:00.0", uuid=f"GPU-{r:032x}", reads=10, timed_seconds=1.0,
                  identity_source="pci_bus_id+uuid") for r in range(4)]
    s = shard.summarize_ranks(named)
    assert s["distinct_devices"] == 4 and s["identity_warning"] is None and s["identity_sources"] == ["pci_bus_id+uuid"]
    # same PCI address, different UUIDs (two partitions of one card): distinct -- and the reverse
    twins = [dict(named[0], rank=0), dict(named[0], rank=1, uuid="GPU-" + "f" * 32)]
    assert shard.summarize_ranks(twins)["distinct_devices"] == 2
    blind = [dict(rank=r, host="n", device_index=0, pci_bus_id=None, uuid=None, reads=10, timed_seconds=1.0) for r in range(2)]
    s = shard.summarize_ranks(blind + [dict(named[3], rank=2)])
    assert s["distinct_devices"] is None and "[0, 1]" in s["identity_warning"] and s["identity_sources"] == ["device_index", "pci_bus_id+uuid"]
    assert [r["identity_source"] for r in s["ranks"]] == ["device_index", "device_index", "pci_bus_id+uuid"]
    # device_identity itself: no GPU here, so the property query fails -- reported, not swallowed
    ident = shard.device_identity(0)
    assert ident["device_index"] == 0 and ident["identity_source"] in ("device_index", "pci_bus_id", "uuid", "pci_bus_id+uuid")
    if ident["identity_source"] == "device_index":
        assert ident["identity_error"]


def test_roofline_traffic_is_tied_to_the_profiled_library(tmp_path):
    """profiles/hbm_traffic.json names the sha256 of the library its PMC passes profiled; bench.py reports
    `roofline.traffic` only for the library it loaded itself, and nulls it with traffic_stale otherwise."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    sha = "ab" * 32
    good = dict(lib_sha256=sha, commit="abc1234", bytes_per_launch=5.0e8, bytes_per_forward=4.0e9, algorithmic_bytes_per_forward=2.2e8,
                bytes_per_forward_by_kernel={"readconv_kernel": 5.0e8})
    path = str(tmp_path / "hbm_traffic.json")
    json.dump(good, open(path, "w"))
    traffic, forward, prov = bench.committed_traffic(path, sha)
    assert traffic == 5.0e8 and forward["bytes"] == 4.0e9 and prov["traffic_stale"] is False and prov["profiled_commit"] == "abc1234"
    # a different library (the kernel changed, the passes were not re-run): no bytes, flagged
    traffic, forward, prov = bench.committed_traffic(path, "cd" * 32)
    assert traffic is None and forward is None and prov["traffic_stale"] is True and prov["profiled_lib_sha256"] == sha
    # a traffic file from before the hash existed is stale by definition
    json.dump({k: v for k, v in good.items() if k != "lib_sha256"}, open(path, "w"))
    assert bench.committed_traffic(path, sha)[2]["traffic_stale"] is True
    # no file: nothing to report, nothing to call stale
    assert bench.committed_traffic(str(tmp_path / "absent.json"), sha) == (None, None, dict(prov, lib_sha256=sha, profiled_lib_sha256=None,
                                                                                          profiled_commit=None, traffic_stale=None))
    # the hash bench.py takes is the hash of the file the engine opens
    import hashlib
    lib = os.path.join(ROOT, "hello_amd", "libhello_mi355x.so")
    if os.path.exists(lib):
        assert bench.lib_sha256() == hashlib.sha256(open(lib, "rb").read()).hexdigest()


def test_bench_launcher_rewrites_only_the_gpus_flag(monkeypatch):
    """`python bench.py --gpus N` from a plain shell starts its ranks as a CHILD torch.distributed.run (never an
    exec, before anything touches a GPU) and returns the child's exit code."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.launch_ranks(8, ["--steps", "5", "--gpus", "8", "--warmup", "2", "--gpus=8", "--no-secondary"])
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "5", "--warmup", "2", "--no-secondary"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    # and main() takes that road only from a plain shell: under a launcher (RANK / WORLD_SIZE set) it never re-launches
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src
