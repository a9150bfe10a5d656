"""BASELINE.json's full-size configuration (single-tech Illumina 30x, 1 M candidate sites on one GPU) through
properties that do not need a million-site oracle run -- every pass over a batch reproduces its first result
bit for bit, site order does not matter beyond float re-association, device memory does not grow along the
stream -- and ONE full launch of every BASELINE.json configuration checked against the CPU oracle on EVERY site."""
import numpy as np
import pytest

from hello_amd import netspec as ns, synth, weights

pytestmark = pytest.mark.gpu

SITES_PER_BATCH = 8192
TOTAL_SITES = 1_000_000


def _sub_batch(batch, sites):
    parts = [batch.site_slice(int(s), int(s) + 1) for s in sites]
    second = batch.reads1 is not None
    return synth.SiteBatch(
        np.concatenate([p.reads0 for p in parts]), np.concatenate([p.reads_per_allele0 for p in parts]),
        np.concatenate([p.alleles_per_site for p in parts]), np.concatenate([p.ref_onehot for p in parts]),
        np.concatenate([p.reads1 for p in parts]) if second else None,
        np.concatenate([p.reads_per_allele1 for p in parts]) if second else None)


def test_million_site_stream_is_reproducible():
    import torch
    from hello_amd.engine import Engine
    from hello_amd.pipeline import HostPipeline
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=31)
    eng = Engine(spec, state, device=0)
    pool = [synth.make_sites(SITES_PER_BATCH, seed=900 + i, coverage=30) for i in range(3)]
    pinned = [synth.SiteBatch(torch.from_numpy(b.reads0).pin_memory(), b.reads_per_allele0, b.alleles_per_site,
                              b.ref_onehot) for b in pool]
    pipe = HostPipeline(eng, depth=2)
    n_steps = -(-TOTAL_SITES // SITES_PER_BATCH)
    first, sites_done, mem = {}, 0, None
    for i in range(n_steps + 2):
        done = pipe.submit(pinned[i % 3], tag=i) if i < n_steps else pipe.flush()
        for tag, logits, meta, post in done:
            k = tag % 3
            if k not in first:
                first[k] = (logits, post)
            else:
                assert np.array_equal(logits, first[k][0]) and np.array_equal(post, first[k][1]), f"step {tag} drifted"
            sites_done += SITES_PER_BATCH
        if i == 8:
            mem = torch.cuda.mem_get_info()[0]       # free bytes on the card: torch's and the engine's allocations
        if i >= n_steps:
            break
    assert sites_done >= TOTAL_SITES
    assert mem - torch.cuda.mem_get_info()[0] < (32 << 20), "device memory grew along the stream"

    # (every site of a full launch against the oracle: test_every_site_of_a_full_launch_matches_the_oracle below)
    assert np.isfinite(first[0][1]).all() and first[0][1].min() >= 0.0 and first[0][1].max() <= 1.0

    # reversing the site order only re-associates the per-allele read sums
    order = np.arange(SITES_PER_BATCH)[::-1]
    rev = _sub_batch(pool[1], order)
    lg_rev, _ = eng.forward_batch(rev)
    aoff1 = np.concatenate([[0], np.cumsum(pool[1].alleles_per_site)])
    back = np.concatenate([first[1][0][0, aoff1[s]:aoff1[s + 1]] for s in order])
    np.testing.assert_allclose(lg_rev[0], back, rtol=1e-5, atol=1e-5)
    eng.close()


FULL_LAUNCHES = [
    # label, model, generator arguments, sites -- the five BASELINE.json configurations (+ hybrid_full: three experts + meta) at the
    # bench's 8 192-site launch; the model variants at 4 096 (their oracle costs 2-4 x as much per site)
    ("C2 Illumina 30x single-tech", "single_tech", dict(coverage=30), SITES_PER_BATCH),
    ("C3 PacBio HiFi, cov U{8..52}, R <= 128", "single_tech", dict(coverage=(8, 52), tech="pacbio", max_reads=128), SITES_PER_BATCH),
    ("C4 hybrid no-ensemble, 30x + 15x", "hybrid_no_ensemble", dict(coverage=30, hybrid_coverage=15), SITES_PER_BATCH),
    ("C5 haplotagged (7 channels), cov U{20..80}", "single_tech_hp", dict(coverage=(20, 80), channels=7, tech="pacbio"), SITES_PER_BATCH),
    ("hybrid full (three experts + meta), 30x + 15x", "hybrid_full", dict(coverage=30, hybrid_coverage=15), SITES_PER_BATCH),
    ("hybrid no-ensemble wide (2x channels: readconv_wide_kernel), 30x + 15x", "hybrid_no_ensemble_wide",
     dict(coverage=30, hybrid_coverage=15), SITES_PER_BATCH // 2),
    ("MoEMergedAdvanced 250 bp feature map (250 bp geometry of the fused kernel, BatchNorm, grouped combiner), 30x + 15x",
     "merged_hybrid_250", dict(coverage=30, hybrid_coverage=15, window=250), SITES_PER_BATCH // 2),
    ("single-tech Softplus / no normalisation (..._layer_norm.py as shipped)", "single_tech_softplus", dict(coverage=30), SITES_PER_BATCH // 2),
]


def _record(line):
    """Append one line to gpurun_out/r06_full_launch_parity.txt (the box merges gpurun_out/ back; kept under profiles/)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "r06_full_launch_parity.txt"), "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass
    print(line)


@pytest.mark.parametrize("label,cfg,kw,n_sites", FULL_LAUNCHES, ids=["C2", "C3", "C4", "C5", "hybrid_full", "wide", "merged_250", "softplus"])
def test_every_site_of_a_full_launch_matches_the_oracle(label, cfg, kw, n_sites, tmp_path, suite_arithmetic):
    """ONE full launch of each BASELINE.json configuration (alleles straddling the fused kernel's read groups,
    workgroup seams and partial slots in both technologies) checked on EVERY site: per-allele sigmoid(logit) of every expert, the
    meta weights and every genotype-pair posterior within 1e-4 of the oracle run one site per call (MixtureOfExpertsAdvanced.py:161-252,
    520-589) -- on a worker pool forked by a child process that never touches the GPU (tests/oracle_pool.py), while the GPU scores
    the same sites.  Also: two runs bit-identical, reversed site order equal up to float re-association, posteriors inside [0, 1].
    Prints and records the worst site with its (reads, alleles)."""
    import time
    from hello_amd.engine import Engine
    from tests import oracle_pool
    out = str(tmp_path / "answers.npz")
    proc = oracle_pool.start(cfg, 33, n_sites, 700 + len(cfg), kw, out)
    try:
        spec = ns.build(cfg)
        state = weights.synth_state(spec, seed=33)
        batch = synth.make_sites(n_sites, seed=700 + len(cfg), **kw)
        eng = Engine(spec, state, device=0)
        t0 = time.perf_counter()
        logits, meta, post = eng.forward_batch(batch, posteriors=True)
        gpu_s = time.perf_counter() - t0
        again, meta2, post2 = eng.forward_batch(batch, posteriors=True)
        assert np.array_equal(logits, again) and np.array_equal(post, post2) and (meta is None or np.array_equal(meta, meta2))
        assert np.isfinite(logits).all() and post.min() >= 0.0 and post.max() <= 1.0 + 1e-6
        order = np.arange(n_sites)[::-1]
        lg_rev, _ = eng.forward_batch(_sub_batch(batch, order))
        aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
        back = np.concatenate([logits[:, aoff[s]:aoff[s + 1]] for s in order], axis=1)
        np.testing.assert_allclose(lg_rev, back, rtol=1e-5, atol=1e-5)
        arithmetic = eng.program.arithmetic
        eng.close()
        want = oracle_pool.collect(proc, out)
    finally:
        if proc.poll() is None:
            proc.kill()
    assert want["logits"].shape == logits.shape and want["post"].shape == post.shape
    probs = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
    d_prob = np.abs(probs - want["probs"]).max(axis=0)                          # per allele, worst expert
    rows = 4 if logits.shape[0] > 1 else 1                                     # single-expert models: the mixture row IS expert 0's
    d_post = np.abs(post[:rows].astype(np.float64) - want["post"][:rows]).max(axis=0)      # per pair
    a = np.asarray(batch.alleles_per_site, np.int64)
    poff = np.concatenate([[0], np.cumsum(a * (a + 1) // 2)])
    per_site = np.maximum(np.maximum.reduceat(d_prob, aoff[:-1]), np.maximum.reduceat(d_post, poff[:-1]))
    d_meta = 0.0
    if meta is not None:
        per_site = np.maximum(per_site, np.abs(meta - want["meta"]).max(axis=1))
        d_meta = float(np.abs(meta - want["meta"]).max())
    # "identical calls" on every site: the called genotype is the most probable pair of the mixture row (caller_calling.py:698-712);
    # a site whose two best pairs tie within 1e-6 in the oracle may legitimately pick either
    got_best = np.array([int(np.argmax(post[0, poff[s]:poff[s + 1]])) for s in range(n_sites)])
    want_rows = [want["post"][0, poff[s]:poff[s + 1]] for s in range(n_sites)]
    same_call = np.array([g == int(np.argmax(w)) or w[g] >= w.max() - 1e-6 for g, w in zip(got_best, want_rows)])
    exact_same = int(sum(g == int(np.argmax(w)) for g, w in zip(got_best, want_rows)))
    worst = int(np.argmax(per_site))
    reads = np.add.reduceat(batch.reads_per_allele0, aoff[:-1]) + (0 if batch.reads1 is None else np.add.reduceat(batch.reads_per_allele1, aoff[:-1]))
    scale = float(np.abs(want["logits"]).max())
    d_logit = float(np.abs(logits - want["logits"]).max())
    _record(f"{label}: {n_sites} sites / {logits.shape[1]} alleles / {int(reads.sum())} reads, arithmetic {arithmetic}: "
            f"max|d sigmoid(logit)| {d_prob.max():.3e}, max|d pair posterior| {d_post.max():.3e}, max|d meta| {d_meta:.3e}, "
            f"max|d logit| {d_logit:.3e} (max |logit| {scale:.1f}); worst site {worst} (R = {int(reads[worst])}, A = {int(a[worst])}): "
            f"{per_site[worst]:.3e}; sites above 1e-5: {int((per_site > 1e-5).sum())}; identical calls {exact_same} / {n_sites} (+ {int(same_call.sum()) - exact_same} ties); oracle {want['seconds']:.1f} s on {want['workers']} "
            f"workers (one site per call), GPU launch {gpu_s * 1e3:.0f} ms host to host")
    assert d_prob.max() < 1e-4 and d_post.max() < 1e-4 and d_meta < 1e-4, (label, worst, int(reads[worst]), int(a[worst]))
    assert same_call.all(), (label, np.nonzero(~same_call)[0][:5])
    np.testing.assert_allclose(logits, want["logits"], rtol=2e-5, atol=2e-4)


def test_read_order_inside_an_allele_only_reassociates_its_sum():
    """reduceSlots (MixtureOfExpertsAdvanced.py:23-34) SUMS an allele's reads: shuffling the reads inside every allele of a full
    8 192-site launch (so that they fall into other read groups, workgroups and partial slots of the fused kernel) moves no logit
    beyond float re-association, and shuffling whole alleles' read blocks with their counts permutes the logits exactly that way."""
    from hello_amd.engine import Engine
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=35)
    eng = Engine(spec, state, device=0)
    batch = synth.make_sites(SITES_PER_BATCH, seed=811, coverage=30)
    logits, _, post = eng.forward_batch(batch, posteriors=True)
    rng = np.random.default_rng(4)
    allele_of_read = np.repeat(np.arange(batch.n_alleles), batch.reads_per_allele0)
    order = np.lexsort((rng.random(allele_of_read.shape[0]), allele_of_read))          # reads stay in their allele, in random order
    assert not np.array_equal(order, np.arange(order.shape[0]))
    shuffled = synth.SiteBatch(batch.reads0[order], batch.reads_per_allele0, batch.alleles_per_site, batch.ref_onehot)
    lg2, _, post2 = eng.forward_batch(shuffled, posteriors=True)
    scale = max(1.0, float(np.abs(logits).max()))
    assert np.abs(lg2 - logits).max() <= 1e-5 * scale and np.abs(post2 - post).max() < 1e-5
    eng.close()
