"""bench.py's configuration table (BASELINE.json configs C2-C5 + hybrid_full), its work accounting, the oracle's reference answers
and the gather's self-check -- the host logic, on the CPU (the GPU legs themselves: tests/test_gpu_pipeline.py)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hello_amd import compiler, netspec as ns, synth, weights     # noqa: E402


def test_config_table_names_the_baseline_configurations_as_the_fullsize_tests_build_them():
    """The table is BASELINE.json's `configs` (SURVEY.md 8d "Concrete configs"): C2 the single-tech 30x headline, C3 PacBio coverage
    U{8..52} with <= 128 reads, C4 the hybrid no-ensemble model with a second read set, C5 seven channels at coverage U{20..80}."""
    import bench
    assert set(bench.BENCH_CONFIGS) == {"C2", "C3", "C4", "C5", "hybrid_full"} and set(bench.SECONDARY_CONFIGS) == set(bench.BENCH_CONFIGS) - {"C2"}
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == 5                                              # C1 (CPU plumbing) + C2..C5
    got = {k: bench.make_config_sites(k, 64, 5) for k in bench.BENCH_CONFIGS}
    # the headline's generator call is unchanged: the seeds of rounds 1-5 give the same pool
    want = synth.make_sites(64, seed=5, coverage=30)
    assert np.array_equal(got["C2"].reads0, want.reads0) and np.array_equal(got["C2"].reads_per_allele0, want.reads_per_allele0)
    assert got["C2"].reads1 is None and got["C3"].reads1 is None and got["C5"].reads1 is None
    assert got["C3"].reads0.shape[2] == 6 and int(np.max(np.add.reduceat(got["C3"].reads_per_allele0, np.concatenate([[0], np.cumsum(got["C3"].alleles_per_site)])[:-1]))) <= 128 + 3
    assert got["C4"].reads1 is not None and got["C4"].reads1.shape[2] == 6 and got["hybrid_full"].reads1 is not None
    assert got["C5"].reads0.shape[2] == 7
    for k, c in bench.BENCH_CONFIGS.items():
        spec = ns.build(c["spec"])
        assert spec.nets["read_convolver0"][0].cin == got[k].reads0.shape[2], k
        assert bool(spec.has("read_convolver1")) == (got[k].reads1 is not None), k


def test_program_flops_are_the_surveys_per_read_per_allele_per_site_figures():
    """per read 10.152 MFLOP (C = 6) / 10.166 (C = 7); per allele 30.966 MFLOP (compressor + xattn); hybrid no-ensemble
    F = 10.152 (R0 + R1) + (2 x 10.322 + 16.515 + 20.644) A + 16.515 S.  bench.py prices a batch from the compiled program's ops."""
    import bench
    for name, per_read, per_allele, per_site in (("C2", 2 * 5_076_096, 2 * (5_160_960 + 10_322_176), 0),
                                                 ("C5", 2 * 5_083_200, 2 * (5_160_960 + 10_322_176), 0),
                                                 ("C4", 2 * 5_076_096, 2 * (2 * 5_160_960 + 8_257_536 + 10_322_176), 2 * 8_257_536)):
        spec = ns.build(bench.BENCH_CONFIGS[name]["spec"])
        program = compiler.compile_model(spec, weights.synth_state(spec, seed=1))
        b = bench.make_config_sites(name, 32, 3)
        reads = b.reads0.shape[0] + (0 if b.reads1 is None else b.reads1.shape[0])
        alg, exe = bench.program_flops(program, b)
        assert alg == per_read * reads + per_allele * b.n_alleles + per_site * b.n_sites, name
        assert 0.5 * alg < exe < alg                                              # Winograd forms execute fewer MFMA FLOPs
        assert bench.batch_input_bytes(b) == b.reads0.size + (0 if b.reads1 is None else b.reads1.size)


def test_checksum_is_a_function_of_the_bytes():
    import bench
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 41)).astype(np.float32)
    a, b = bench.checksum(x), bench.checksum(x.copy())
    assert a == b and a["n"] == 123 and a["first"] == float(x[0, 0]) and a["last"] == float(x[-1, -1])
    y = x.copy()
    y[1, 7] = np.nextafter(y[1, 7], np.float32(10))
    assert bench.checksum(y) != a and bench.checksum(y)["crc32"] != a["crc32"]
    assert bench.checksum(x[:, ::-1])["crc32"] != a["crc32"]                      # order matters: a permuted gather is caught
    empty = bench.checksum(np.zeros((1, 0), np.float32))
    assert empty["n"] == 0 and empty["first"] is None and empty["sum"] == 0.0
    # survives the trip the reports make (pickle through all_gather_object; JSON on the line)
    assert json.loads(json.dumps(a)) == a


def test_feature_dicts_are_the_per_site_call_arguments():
    """caller_calling.py:631-649: {allele: (float [R, L, C], float [R', L, C] | None)} in allele order + a [1, L, 5] segment."""
    import bench
    for name in ("C2", "C4"):
        b = bench.make_config_sites(name, 6, 9)
        calls = bench.feature_dicts(b)
        assert len(calls) == 6
        roff0 = np.concatenate([[0], np.cumsum(b.reads_per_allele0)])
        a = 0
        for s, (fd, seg) in enumerate(calls):
            assert seg.shape == (1, 150, 5) and seg.dtype == np.float32 and len(fd) == int(b.alleles_per_site[s])
            for key, (first, second) in fd.items():
                assert first.dtype == np.float32 and np.array_equal(first, b.reads0[roff0[a]:roff0[a + 1]])
                assert (second is None) == (b.reads1 is None)
                if second is not None:
                    assert second.shape[0] == int(b.reads_per_allele1[a])
                a += 1


def test_oracle_answers_are_the_per_site_oracle_on_the_check_sites():
    """The reference answers of bench.py's parity legs: a forked pool of per-site oracle calls, stitched back in site order; equal to
    the oracle called directly, for a single-expert and for the three-expert + meta model."""
    import bench
    from oracle import moe_oracle as mo
    got = bench.oracle_answers(["C2", "hybrid_full"], seed=2, n_sites=10)
    for name in ("C2", "hybrid_full"):
        check, probs, meta, post = got[name]
        spec = ns.build(bench.BENCH_CONFIGS[name]["spec"])
        oracle = mo.Oracle(spec, weights.synth_state(spec, seed=2))
        want_logits, want_meta = mo.forward_batch(oracle, check, chunk_sites=1)
        assert check.n_sites == 10 and np.array_equal(check.reads0, bench.make_config_sites(name, 10, 2 + 4242).reads0)
        np.testing.assert_allclose(probs, mo.sigmoid(want_logits), rtol=0, atol=1e-7)
        assert probs.shape == (3 if name == "hybrid_full" else 1, check.n_alleles)
        assert (meta is None) == (want_meta is None)
        a = np.asarray(check.alleles_per_site, np.int64)
        assert post.shape == (4, int((a * (a + 1) // 2).sum())) and np.isfinite(post).all() and post.min() >= 0 and post.max() <= 1 + 1e-6
        if name == "C2":                                                          # single expert: the mixture row is expert 0's
            assert np.array_equal(post[0], post[1])
        else:
            np.testing.assert_allclose(meta, want_meta, atol=1e-7)
            np.testing.assert_allclose(meta.sum(axis=1), 1.0, atol=1e-6)


def test_lib_sha256_follows_the_library_the_engine_opens(tmp_path, monkeypatch):
    """the hash on the line is the hash of the file hello_amd.engine OPENED (HELLO_LIB under a kernel A/B run), and an
    unreadable file gives None (traffic then reported stale) instead of losing the measurement after the run."""
    import bench
    from hello_amd import engine
    other = tmp_path / "libother.so"
    other.write_bytes(b"not the in-tree library")
    monkeypatch.setattr(engine, "_LIB_PATH", str(other))
    assert bench.lib_sha256() == hashlib.sha256(b"not the in-tree library").hexdigest()
    monkeypatch.setattr(engine, "_LIB_PATH", str(tmp_path / "absent.so"))
    assert bench.lib_sha256() is None
    assert bench.committed_traffic(os.path.join(ROOT, "profiles", "hbm_traffic.json"), None)[2]["traffic_stale"] in (True, None)


def test_oracle_pool_answers_equal_the_oracle_called_directly(tmp_path):
    """tests/oracle_pool.py (the child process tree behind the GPU suite's exhaustive full-launch parity): started as a child, it
    regenerates the batch from its seed, scores every site through the per-site oracle on a forked pool and writes probabilities,
    meta weights and pair posteriors in site order -- equal to the oracle called directly, for a hybrid ensemble model."""
    from oracle import moe_oracle as mo
    from tests import oracle_pool
    kw = dict(coverage=12, hybrid_coverage=6)
    out = str(tmp_path / "answers.npz")
    got = oracle_pool.collect(oracle_pool.start("hybrid_full", 33, 14, 77, kw, out), out, timeout=600)
    spec = ns.build("hybrid_full")
    batch = synth.make_sites(14, seed=77, **kw)
    want_logits, want_meta = mo.forward_batch(mo.Oracle(spec, weights.synth_state(spec, seed=33), backend="torch"), batch, chunk_sites=1)
    assert got["logits"].shape == (3, batch.n_alleles) and got["workers"] >= 1 and got["seconds"] > 0
    np.testing.assert_allclose(got["logits"], want_logits, rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["probs"], mo.sigmoid(want_logits), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["meta"], want_meta, rtol=0, atol=1e-6)
    a = np.asarray(batch.alleles_per_site, np.int64)
    p_off = np.concatenate([[0], np.cumsum(a * (a + 1) // 2)])
    a_off = np.concatenate([[0], np.cumsum(a)])
    assert got["post"].shape == (4, int(p_off[-1]))
    for s in (0, 5, 13):
        rows = mo.posteriors([mo.sigmoid(want_logits[e, a_off[s]:a_off[s + 1]]) for e in range(3)], want_meta[s])
        np.testing.assert_allclose(got["post"][:, p_off[s]:p_off[s + 1]], np.stack(rows), rtol=0, atol=1e-6)
