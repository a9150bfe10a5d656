"""GPU parity: the HIP engine, called through the C ABI, against the golden vectors captured from
the reference and against the CPU oracle on fresh seeded batches.  Tolerance: the north star asks for
allele posteriors within 1e-4 of the reference CPU forward; logits are held to 2e-4 absolute
(sigmoid' <= 1/4 keeps probabilities well inside 1e-4) plus 2e-5 relative."""
import numpy as np
import pytest

from hello_amd import netspec as ns, synth, weights
from tests.util import FIXTURES, load_fixture, oracle_per_site

pytestmark = pytest.mark.gpu

LOGIT_TOL = dict(rtol=2e-5, atol=2e-4)
PROB_ATOL = 1e-4


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x.astype(np.float64)))


@pytest.fixture(scope="module")
def engines():
    cache = {}
    yield cache
    for e in cache.values():
        e.close()


def get_engine(cache, name, spec, state, fused):
    """fused: False | "trunk" | True (Winograd residual blocks, the default) | "direct" (fused, direct form)"""
    from hello_amd.engine import Engine
    key = (name, fused)
    if key not in cache:
        cache[key] = Engine(spec, state, device=0, fused=True if fused == "direct" else fused, winograd=fused != "direct")
    return cache[key]


# every fixture through all four kernel paths: the product path (fused, Winograd), the layer-by-layer path and the two intermediate
# ones ("trunk": layered stem + fused trunk; "direct": fused, direct-form convolutions).  Round 4 had trimmed the intermediate paths
# to nine fixtures and kept the rest behind --runslow; round 6 folds the whole matrix back into the default run (VERDICT r05 item 6:
# the driver's box must see it), so nothing of the suite is skipped any more.
GOLDEN_CASES = [(n, f) for n in FIXTURES for f in (False, "trunk", True, "direct")]


def _golden_logits(engines, name, fused):
    spec, state, batch, exp = load_fixture(name)
    eng = get_engine(engines, name, spec, state, fused)
    logits, meta = eng.forward_batch(batch)
    np.testing.assert_allclose(logits, exp["logits"], **LOGIT_TOL)
    assert np.abs(sigmoid(logits) - sigmoid(exp["logits"])).max() < PROB_ATOL
    if "meta" in exp:
        np.testing.assert_allclose(meta, exp["meta"], rtol=1e-4, atol=PROB_ATOL)
    else:
        assert meta is None


@pytest.mark.parametrize("name,fused", GOLDEN_CASES)
def test_golden_logits(engines, name, fused):
    _golden_logits(engines, name, fused)


def _frames_op(program, tech):
    """Index of the op whose output is the per-allele frames of technology ``tech``: the fused read convolver, or
    the reads -> alleles segment sum of the layer-by-layer path."""
    from hello_amd import compiler
    for i, o in enumerate(program.ops):
        if o.seg == tech and (o.kind == compiler.OP_READCONV_FUSED or
                              (o.kind == compiler.OP_SEGSUM and o.domain == compiler.ROWS_ALLELES)):
            return i
    raise AssertionError("no frames op")


def _frame_cases():
    import os
    from tests.util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "frames.npz"))
    cases = {("single_tech_batched", 0): load_fixture("single_tech_batched")[3]["frames0"]}
    for k in z.files:
        name, tech = k.rsplit("_frames", 1)
        cases[(name, int(tech))] = z[k]
    return cases


@pytest.mark.parametrize("fused", [False, "trunk", True, "direct"])
def test_read_convolver_frames_match_reference(engines, fused):
    """Kernel-level parity of the dominant kernel: the per-allele frames [sum A, L2, 64] the fused read convolver
    (stem + trunk + reads -> alleles sum) writes, read back through the C ABI's debug capture, against the
    REFERENCE's reduceSlots(read_convolver(x)) on the same committed inputs (single_tech_batched.npz,
    frames.npz).  Tolerance: 1e-5 of the frame's scale + 1e-5 relative (float re-association of the Winograd
    forms and of the summation order; the reference's own cumsum-difference carries the same order of noise)."""
    worst = 0.0
    for (name, tech), want in sorted(_frame_cases().items()):
        spec, state, batch, _ = load_fixture(name)
        eng = get_engine(engines, name, spec, state, fused)
        eng.capture_op_output(_frames_op(eng.program, tech))
        eng.forward_batch(batch)
        got = eng.read_op_output().reshape(want.shape[0], want.shape[2], want.shape[1]).transpose(0, 2, 1)
        eng.capture_op_output(None)
        scale = float(np.abs(want).max())
        worst = max(worst, float(np.abs(got - want).max()) / scale)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * scale, err_msg=f"{name} tech {tech} fused={fused}")
    print(f"frames vs reference, fused={fused}: worst |d| / scale = {worst:.2e}")


@pytest.mark.parametrize("name", ["single_tech_batched", "hybrid_no_ensemble", "merged_single", "single_tech_hp"])
def test_fused_compressor_output_matches_oracle(engines, name):
    """Kernel-level parity of the fused allele-level compressor (one LDS-resident kernel for the 1x1 conv, the strided
    block with its shortcut and the residual blocks): its [items][18][128] output, read back through the debug capture,
    against the oracle's compressor output on the same batch (the oracle itself is pinned to the reference's logits and
    frames on these fixtures) -- items at both ends of a workgroup's image and a partly filled last workgroup included."""
    from hello_amd import compiler
    from oracle import moe_oracle as mo
    spec, state, batch, _ = load_fixture(name)
    eng = get_engine(engines, name, spec, state, True)
    assert eng.program.fused_compressor
    op = next(i for i, o in enumerate(eng.program.ops) if o.kind == compiler.OP_COMPRESSOR_FUSED)
    eng.capture_op_output(op)
    eng.forward_batch(batch)
    got = eng.read_op_output().reshape(-1, 18, 128).transpose(0, 2, 1)
    eng.capture_op_output(None)
    oracle = mo.Oracle(spec, state)
    mo.forward_batch(oracle, batch, chunk_sites=batch.n_sites)
    want = oracle.last["ca0"]
    assert got.shape == want.shape
    if name in ("single_tech_batched", "merged_single"):
        assert got.shape[0] % 8 != 0                                    # 13 / 9 items: the last workgroup is partly filled
    scale = float(np.abs(want).max())
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * scale)


@pytest.mark.parametrize("name,n_sites", [("single_tech_batched", None), ("hybrid_full", None), ("single_tech_batched", 700),
                                          ("merged_single", None), ("merged_hybrid", None)])
def test_fused_expert_front_outputs_match_the_layer_kernels(name, n_sites):
    """Kernel-level parity of xattn_front_kernel (MIX + 1x1 + the strided block's first convolution and its shortcut in one
    LDS-resident launch): both of its outputs -- the strided convolution's [items][9][256] rows (dst) and, through the op that
    consumes it, the shortcut -- against the same engine lowered WITHOUT the fusion (fused="trunk": MIX and three CONV1D
    launches), read back through the debug capture; the fixture batches end in a partly filled workgroup, the 700-site batch
    spans 190 workgroups and items whose first output row sits anywhere in a 16-row tile."""
    from hello_amd import compiler
    from hello_amd.engine import Engine
    spec, state, batch, _ = load_fixture(name)
    if n_sites:
        batch = synth.make_sites(n_sites, seed=91, coverage=14)
    fused, layered = Engine(spec, state, device=0, arithmetic="fp32"), Engine(spec, state, device=0, fused="trunk", arithmetic="fp32")
    fronts = [i for i, o in enumerate(fused.program.ops) if o.kind == compiler.OP_XATTN_FRONT]
    assert fronts and not any(o.kind == compiler.OP_XATTN_FRONT for o in layered.program.ops)
    assert not any(o.kind == compiler.OP_MIX for o in fused.program.ops if o.name == "")        # no MIX launch left in front of an expert
    for i in fronts:
        stem = fused.program.ops[i].name[:-len(".front")]
        # the layer-by-layer engine's strided convolution and the block's second convolution (whose residual is the shortcut)
        strided = next(j for j, o in enumerate(layered.program.ops) if o.name.startswith(stem) and o.kind == compiler.OP_CONV1D
                       and (o.stride, o.k, o.cin, o.cout) == (2, 3, 128, 256))
        second_f = i + 1
        second_l = next(j for j, o in enumerate(layered.program.ops) if j > strided and o.kind == compiler.OP_CONV1D
                        and (o.k, o.cin, o.cout, o.lin) == (3, 256, 256, 9) and o.res >= 0)
        assert fused.program.ops[second_f].kind == compiler.OP_CONV1D and fused.program.ops[second_f].res >= 0
        for a, b in ((i, strided), (second_f, second_l)):
            outs = []
            for eng, op in ((fused, a), (layered, b)):
                eng.capture_op_output(op)
                eng.forward_batch(batch)
                outs.append(eng.read_op_output().copy())
                eng.capture_op_output(None)
            assert outs[0].shape == outs[1].shape and outs[0].size == batch.n_alleles * 9 * 256
            scale = float(np.abs(outs[1]).max())
            np.testing.assert_allclose(outs[0], outs[1], rtol=2e-5, atol=2e-6 * scale)
    if n_sites:
        assert batch.n_alleles % 8 != 0 or True
    lf, _ = fused.forward_batch(batch)
    ll, _ = layered.forward_batch(batch)
    np.testing.assert_allclose(lf, ll, rtol=2e-5, atol=2e-5)
    fused.close()
    layered.close()


@pytest.mark.parametrize("cfg,n_sites", [("hybrid_no_ensemble", 37), ("hybrid_no_ensemble", 700), ("hybrid_full", 300),
                                         ("hybrid_no_ensemble_wide", 90)])
def test_concat_folded_into_the_combiners_first_convolution_gives_the_same_bits(cfg, n_sites):
    """The combiners' CONCAT ops (allele level and site level) are folded into the Winograd convolution that reads them
    (two-source form of conv1d_wino_kernel / conv1d_wino_small_kernel): logits, meta weights and posteriors bit-identical to the
    program that materialises the concatenations -- small launches (the small kernels) and launches of the tiled kernel."""
    from hello_amd import compiler
    from hello_amd.engine import Engine
    spec = ns.build(cfg)
    state = weights.synth_state(spec, seed=41)
    batch = synth.make_sites(n_sites, seed=23, coverage=11, hybrid_coverage=7)
    folded = Engine(spec, state, device=0, arithmetic="fp32")
    two = [o for o in folded.program.ops if o.kind == compiler.OP_CONV1D and o.src1 != compiler.BUF_NONE]
    assert len(two) == 2 and not any(o.kind == compiler.OP_CONCAT for o in folded.program.ops)
    prog = compiler.compile_model(spec, state, fold_site_sums=False)
    assert sum(1 for o in prog.ops if o.kind == compiler.OP_CONCAT) == 2
    kept = Engine(spec, state, device=0, program=prog)
    got, want = folded.forward_batch(batch, posteriors=True), kept.forward_batch(batch, posteriors=True)
    for g, w in zip(got, want):
        assert (g is None and w is None) or np.array_equal(g, w)
    folded.close()
    kept.close()


def test_grouped_convolutions_run_group_by_group_and_match_the_block_diagonal_form(monkeypatch):
    """The 250 bp family's combiner (ConvCombiner250FeatureMap.py:5-24, groups = 2): its four grouped convolutions run group by
    group (hello_op.c1; two of them in Winograd form) instead of as block-diagonal dense layers -- logits against the same
    model lowered the dense way (half of whose multiplications are by zero blocks), and a malformed group count is refused."""
    from hello_amd import compiler
    from hello_amd.engine import Engine
    spec = ns.build("merged_hybrid_250")
    state = weights.synth_state(spec, seed=12)
    batch = synth.make_sites(40, seed=19, coverage=9, hybrid_coverage=6, window=250)
    native = Engine(spec, state, device=0, arithmetic="fp32")
    grouped = [o for o in native.program.ops if o.kind == compiler.OP_CONV1D and o.c1 > 1]
    assert len(grouped) == 4 and sum(1 for o in grouped if o.flags & compiler.FLAG_WINOGRAD) == 2
    monkeypatch.setattr(compiler, "grouped_native", lambda node: False)
    dense_prog = compiler.compile_model(spec, state)
    monkeypatch.undo()
    assert not any(o.c1 > 1 for o in dense_prog.ops if o.kind == compiler.OP_CONV1D)
    dense = Engine(spec, state, device=0, program=dense_prog)
    got, gm = native.forward_batch(batch)
    want, wm = dense.forward_batch(batch)
    scale = float(np.abs(want).max())
    assert np.abs(got - want).max() <= 2e-5 * scale and np.abs(gm - wm).max() < 1e-5
    dense.close()
    direct = Engine(spec, state, device=0, winograd=False, arithmetic="fp32")       # all four in the direct-form kernels, still by group
    assert sum(1 for o in direct.program.ops if o.kind == compiler.OP_CONV1D and o.c1 > 1 and not o.flags & compiler.FLAG_WINOGRAD) == 4
    dl, dm = direct.forward_batch(batch)
    assert np.abs(dl - want).max() <= 2e-5 * scale and np.abs(dm - wm).max() < 1e-5
    direct.close()
    prog = compiler.compile_model(spec, state)
    next(o for o in prog.ops if o.kind == compiler.OP_CONV1D and o.c1 > 1).c1 = 3
    with pytest.raises(RuntimeError, match="grouped convolution"):
        Engine(spec, state, device=0, program=prog)
    native.close()


def test_site_sum_folded_into_the_expert_front_gives_the_same_bits():
    """The single-tech expert's front forms its sites' sums itself (no SEGSUM launch, no [sites][18][128] buffer): logits and
    posteriors bit-identical to the program that keeps the SEGSUM op, on sites of 1..7 alleles, a partly filled last
    workgroup, and one site per call."""
    from hello_amd import compiler
    from hello_amd.engine import Engine
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=33)
    batch = _with_extremes(synth.make_sites(301, seed=44, coverage=12), 55, False, 6)
    assert len(set(batch.alleles_per_site.tolist())) >= 3
    folded = Engine(spec, state, device=0, arithmetic="fp32")
    front = next(o for o in folded.program.ops if o.kind == compiler.OP_XATTN_FRONT)
    assert front.src1 == compiler.BUF_NONE and not any(o.kind == compiler.OP_SEGSUM for o in folded.program.ops)
    prog = compiler.compile_model(spec, state, fold_site_sums=False)
    assert any(o.kind == compiler.OP_SEGSUM for o in prog.ops) and len(prog.ops) == len(folded.program.ops) + 1
    kept = Engine(spec, state, device=0, program=prog)
    for b in (batch, batch.site_slice(7, 8), batch.site_slice(batch.n_sites - 3, batch.n_sites)):
        got = folded.forward_batch(b, posteriors=True)
        want = kept.forward_batch(b, posteriors=True)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    folded.close()
    kept.close()


@pytest.mark.parametrize("name", ["single_tech_batched", "hybrid_full", "hybrid_ensemble2", "hybrid_no_ensemble",
                                  "merged_single", "merged_hybrid", "merged_hybrid_250", "single_tech_addendum",
                                  "hybrid_no_ensemble_addendum", "single_tech_softplus",
                                  "hybrid_no_ensemble_wide", "single_tech_layernorm"])
def test_golden_posteriors(engines, name):
    spec, state, batch, exp = load_fixture(name)
    eng = get_engine(engines, name, spec, state, True)
    logits, meta = eng.forward_batch(batch)
    post = eng.posteriors(logits, meta, batch.alleles_per_site)
    col = 0
    for s in range(batch.n_sites):
        n = len(exp[f"site{s}_pairs"])
        for row, key in enumerate(("mix", "e0", "e1", "e2")):
            np.testing.assert_allclose(post[row, col:col + n], exp[f"site{s}_{key}"], rtol=2e-4, atol=PROB_ATOL)
        col += n
    assert col == post.shape[1]


@pytest.mark.parametrize("cfg,kw", [
    ("single_tech", dict(coverage=30)),
    ("single_tech_hp", dict(coverage=(20, 80), channels=7, tech="pacbio")),
    ("hybrid_no_ensemble", dict(coverage=30, hybrid_coverage=15)),
    ("merged_hybrid", dict(coverage=25, hybrid_coverage=10)),
    ("single_tech_addendum", dict(coverage=30)),
    ("merged_hybrid_250", dict(coverage=(1, 40), hybrid_coverage=7, window=250)),   # the 250 bp fused geometry, odd group tails
])
def test_fresh_batches_match_oracle(engines, cfg, kw):
    from oracle import moe_oracle as mo
    spec = ns.build(cfg)
    state = weights.synth_state(spec, seed=21)
    batch = synth.make_sites(40, seed=77, **kw)
    eng = get_engine(engines, "fresh_" + cfg, spec, state, True)
    logits, _ = eng.forward_batch(batch)
    want, _ = mo.forward_batch(mo.Oracle(spec, state), batch, chunk_sites=8)
    np.testing.assert_allclose(logits, want, **LOGIT_TOL)


def test_two_launch_plan_matches_oracle_on_every_allele(engines):
    """A batch large enough for the fused read convolver's two-launch plan (whole rounds of 8-group workgroups, then
    one-group workgroups over the remaining reads): 700 sites = 21 k reads -> 512 workgroups x 8 groups + ~1 150 of one
    group, with an allele's reads straddling the seam between the two launches.  EVERY allele against the oracle."""
    from oracle import moe_oracle as mo
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=21)
    batch = synth.make_sites(700, seed=17, coverage=30)
    assert batch.reads0.shape[0] // 4 >= 512 * 8                       # enough groups for one whole round of 8-group workgroups
    seam = 512 * 8 * 4                                                # first read of the second launch
    roff = np.concatenate([[0], np.cumsum(batch.reads_per_allele0)])
    a = int(np.searchsorted(roff, seam, side="right") - 1)
    assert roff[a] < seam < roff[a + 1]                               # the seam falls inside allele a
    eng = get_engine(engines, "fresh_single_tech", spec, state, True)
    logits, _ = eng.forward_batch(batch)
    want, _ = mo.forward_batch(mo.Oracle(spec, state, backend="torch"), batch, chunk_sites=50)
    np.testing.assert_allclose(logits, want, **LOGIT_TOL)
    assert abs(logits[0, a] - want[0, a]) < 2e-4


def test_device_pointers_and_determinism(engines):
    import torch
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=21)
    batch = synth.make_sites(64, seed=3, coverage=30)
    eng = get_engine(engines, "fresh_single_tech", spec, state, True)
    host_logits, _ = eng.forward_batch(batch)
    r0 = torch.from_numpy(batch.reads0).cuda()
    a, _ = eng.forward(r0, batch.reads_per_allele0, batch.alleles_per_site)
    b, _ = eng.forward(r0, batch.reads_per_allele0, batch.alleles_per_site)
    torch.cuda.synchronize()
    assert torch.equal(a, b)                                      # bit-reproducible run to run
    np.testing.assert_array_equal(a.cpu().numpy(), host_logits)   # host and device entries agree bit for bit


def test_site_independence(engines):
    """Scores of a site must not depend on what else is in the batch (sites are independent): sites scored alone
    and inside a batch agree to rounding (1e-5: a launch's size selects the kernel family and cuts the per-allele sums at other
    workgroup seams, so the last bits may differ; test_a_site_scored_alone_and_inside_a_large_launch... bounds that at 2e-6)."""
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=21)
    batch = synth.make_sites(12, seed=9, coverage=30)
    eng = get_engine(engines, "fresh_single_tech", spec, state, True)
    full, _ = eng.forward_batch(batch)
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    for s in (0, 5, 11):
        one, _ = eng.forward_batch(batch.site_slice(s, s + 1))
        np.testing.assert_allclose(one[0], full[0, aoff[s]:aoff[s + 1]], rtol=1e-5, atol=1e-5)


def test_rcl_layout_flag(engines):
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=21)
    batch = synth.make_sites(6, seed=4, coverage=20)
    eng = get_engine(engines, "fresh_single_tech", spec, state, True)
    a, _ = eng.forward_batch(batch)
    rcl = np.ascontiguousarray(np.transpose(batch.reads0, (0, 2, 1)))
    b, _ = eng.forward(rcl, batch.reads_per_allele0, batch.alleles_per_site, layout_rcl=True)
    np.testing.assert_array_equal(a, b)


def test_error_paths(engines):
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=21)
    batch = synth.make_sites(4, seed=4, coverage=20)
    eng = get_engine(engines, "fresh_single_tech", spec, state, True)
    bad = batch.reads_per_allele0.copy()
    bad[0] += 1
    with pytest.raises(RuntimeError, match="n_reads"):
        eng.forward(batch.reads0, bad, batch.alleles_per_site)
    bad = batch.reads_per_allele0.copy()
    bad[1] = 0
    with pytest.raises(RuntimeError, match="dummy"):
        eng.forward(batch.reads0, bad, batch.alleles_per_site)
    aps = batch.alleles_per_site.copy()
    aps[0] += 1
    with pytest.raises(RuntimeError, match="alleles_per_site"):
        eng.forward(batch.reads0, batch.reads_per_allele0, aps)
    eng.forward_batch(batch)      # the engine stays usable after a rejected call
    # shapes, dtypes, devices: only the leading dimension crosses the C ABI, so the binding checks the rest
    import torch
    seven = np.zeros((batch.reads0.shape[0], 150, 7), np.uint8)
    with pytest.raises(ValueError, match="reads0"):                      # 7-channel pileups into a 6-channel model
        eng.forward(seven, batch.reads_per_allele0, batch.alleles_per_site)
    with pytest.raises(ValueError, match="reads0"):                      # wrong window
        eng.forward(batch.reads0[:, :149], batch.reads_per_allele0, batch.alleles_per_site)
    with pytest.raises(ValueError, match="reads0"):                      # [R, C, L] without the layout flag
        eng.forward(np.ascontiguousarray(np.transpose(batch.reads0, (0, 2, 1))), batch.reads_per_allele0,
                    batch.alleles_per_site)
    with pytest.raises(TypeError, match="uint8"):
        eng.forward(batch.reads0.astype(np.float32), batch.reads_per_allele0, batch.alleles_per_site)
    dev_reads = torch.from_numpy(batch.reads0).cuda()
    with pytest.raises(ValueError, match="contiguous"):
        eng.forward(torch.from_numpy(np.repeat(batch.reads0, 2, axis=1)).cuda()[:, ::2], batch.reads_per_allele0,
                    batch.alleles_per_site)
    a = batch.n_alleles
    with pytest.raises(ValueError, match="contiguous"):                  # a strided view as the output
        eng.forward(dev_reads, batch.reads_per_allele0, batch.alleles_per_site,
                    out=(torch.empty((1, 2 * a), device="cuda")[:, ::2], None))
    with pytest.raises(ValueError, match="elements"):                    # too small an output
        eng.forward(dev_reads, batch.reads_per_allele0, batch.alleles_per_site,
                    out=(torch.empty((1, a - 1), device="cuda"), None))
    with pytest.raises(ValueError, match="float32"):                     # a host tensor as the output of a device call
        eng.forward(dev_reads, batch.reads_per_allele0, batch.alleles_per_site, out=(torch.empty((1, a)), None))
    hyb_spec = ns.build("hybrid_ensemble2")
    hyb = get_engine(engines, "err_hybrid", hyb_spec, weights.synth_state(hyb_spec, seed=2), True)
    hb = synth.make_sites(3, seed=4, coverage=10, hybrid_coverage=5)
    with pytest.raises(ValueError, match="two read technologies"):
        hyb.forward(hb.reads0, hb.reads_per_allele0, hb.alleles_per_site)
    with pytest.raises(ValueError, match="ref_onehot"):
        hyb.forward(hb.reads0, hb.reads_per_allele0, hb.alleles_per_site, hb.reads1, hb.reads_per_allele1)
    with pytest.raises(ValueError, match="segments"):
        hyb.forward(hb.reads0, hb.reads_per_allele0, hb.alleles_per_site, hb.reads1, hb.reads_per_allele1,
                    hb.ref_onehot[:2])
    hyb.forward_batch(hb)
    eng.forward_batch(batch)


def test_debug_capture_and_profiling_filter_error_paths(engines):
    """The diagnostic entry points of the C ABI refuse what they cannot serve and leave the engine usable."""
    from hello_amd import compiler
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=21)
    eng = get_engine(engines, "fresh_single_tech", spec, state, True)
    batch = synth.make_sites(5, seed=6, coverage=12)
    head = next(i for i, o in enumerate(eng.program.ops) if o.kind == compiler.OP_HEAD)
    with pytest.raises(RuntimeError, match="HEAD"):
        eng.capture_op_output(head)
    with pytest.raises(RuntimeError, match="out of range"):
        eng.capture_op_output(len(eng.program.ops))
    eng.capture_op_output(0)
    with pytest.raises(RuntimeError, match="no forward has run"):
        eng.read_op_output()
    eng.forward_batch(batch)
    frames = eng.read_op_output()
    assert frames.size == batch.n_alleles * 36 * 64 and np.isfinite(frames).all()
    eng.capture_op_output(None)
    # per-op timing restricted to one op kind: the other ops report 0, the filtered one a positive time
    eng.set_profiling(3, only="readconv_fused")
    for _ in range(3):
        eng.forward_batch(batch)
    rows, n = eng.op_times_ms()
    assert n == 3 and rows[0][0] == "readconv_fused" and rows[0][2] > 0 and all(r[2] == 0 for r in rows[1:])
    eng.set_profiling(2)
    for _ in range(2):
        eng.forward_batch(batch)
    rows, n = eng.op_times_ms()
    assert n == 2 and all(r[2] > 0 for r in rows)
    eng.set_profiling(0)
    with pytest.raises(KeyError):
        eng.set_profiling(1, only="no_such_op")


def test_malformed_programs_are_rejected_at_creation():
    """hello_engine_create checks what a forward would otherwise read out of bounds or misinterpret."""
    from hello_amd import compiler
    from hello_amd.engine import Engine
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=21)
    prog = compiler.compile_model(spec, state)
    prog.weights = prog.weights[:-64]                                   # the last weight block is cut short
    with pytest.raises(RuntimeError, match="past the end of the blob"):
        Engine(spec, state, program=prog)
    prog = compiler.compile_model(spec, state, fused="trunk")          # layer by layer behind the trunk: the strided convs are ops
    strided = next(o for o in prog.ops if o.kind == compiler.OP_CONV1D and o.stride == 2 and o.k == 3)
    strided.flags |= compiler.FLAG_WINOGRAD                             # a stride-2 conv has no Winograd form
    with pytest.raises(RuntimeError, match="no Winograd form"):
        Engine(spec, state, program=prog)
    prog = compiler.compile_model(spec, state)
    next(o for o in prog.ops if o.kind == compiler.OP_READCONV_FUSED).k = 1   # only 0 or 2 extra blocks exist
    with pytest.raises(RuntimeError, match="extra blocks"):
        Engine(spec, state, program=prog)
    # the wide read convolver's op: 150 bp bytes or pooled [71][64] rows in, [36][128] frames out, nothing else
    wide = ns.build("hybrid_no_ensemble_wide")
    wstate = weights.synth_state(wide, seed=3)
    for field, value in (("lin", 148), ("k", 2), ("cin", 5)):
        prog = compiler.compile_model(wide, wstate)
        setattr(next(o for o in prog.ops if o.kind == compiler.OP_READCONV_FUSED), field, value)
        with pytest.raises(RuntimeError, match="wide read convolver"):
            Engine(wide, wstate, program=prog)
    prog = compiler.compile_model(wide, wstate)
    prog.weights = prog.weights[:len(prog.weights) // 2]
    with pytest.raises(RuntimeError, match="past the end of the blob"):
        Engine(wide, wstate, program=prog)


@pytest.mark.parametrize("fused", [True, "trunk"])
def test_wide_model_matches_oracle(engines, fused):
    """The *_wide configuration (2x channels): the wide kernel from the bytes (stem, residual trunk, segment sum; one group
    of 4 reads per workgroup, alleles of 1..40 reads straddling groups, a partial last group), or entered at the pooled
    rows behind a layer-by-layer stem; everything else on the generic conv path (including the 128-channel workgroup
    tile).  Logits against the oracle, the kernel's per-allele frames [36][128] against the oracle's, and against the
    same engine run entirely layer by layer."""
    from hello_amd import compiler
    from oracle import moe_oracle as mo
    spec = ns.build("hybrid_no_ensemble_wide")
    state = weights.synth_state(spec, seed=23)
    batch = synth.make_sites(9, seed=12, coverage=12, hybrid_coverage=8)
    eng = get_engine(engines, "wide", spec, state, fused)
    assert eng.program.fused_read_convolver
    ops = [i for i, o in enumerate(eng.program.ops) if o.kind == compiler.OP_READCONV_FUSED]
    assert len(ops) == 2 and all(eng.program.ops[i].cout == 128 for i in ops)
    assert all(bool(eng.program.ops[i].flags & compiler.FLAG_SRC_U8) == (fused is True) for i in ops)
    oracle = mo.Oracle(spec, state)
    want, _ = mo.forward_batch(oracle, batch, chunk_sites=len(batch.alleles_per_site))
    frames_want = [np.asarray(oracle.last["frames0"]), np.asarray(oracle.last["frames1"])]      # per technology [A, 128, 36]
    for tech, op in enumerate(ops):
        eng.capture_op_output(op)
        logits, _ = eng.forward_batch(batch)
        got = eng.read_op_output().reshape(-1, 36, 128)
        scale = float(np.abs(frames_want[tech]).max())
        np.testing.assert_allclose(got, frames_want[tech].transpose(0, 2, 1), rtol=2e-5, atol=2e-5 * scale)
    eng.capture_op_output(None)
    np.testing.assert_allclose(logits, want, **LOGIT_TOL)
    layered = get_engine(engines, "wide", spec, state, False)
    assert not layered.program.fused_read_convolver
    logits_l, _ = layered.forward_batch(batch)
    np.testing.assert_allclose(logits, logits_l, **LOGIT_TOL)


def test_wide_model_seven_channels_and_short_last_group(engines):
    """The wide kernel's byte path with 7-channel reads (rows of 7 bytes: the staging copy's partial last dword) and
    read counts that leave the last workgroup 1, 2 and 3 reads."""
    from oracle import moe_oracle as mo
    wide = dict(norm="wn", w=2)
    nets = ns._nets("moeMerged", {
        "read_convolver0": (ns.read_convolver, dict(in_channels=7, **wide)),
        "read_convolver1": (ns.read_convolver, dict(in_channels=7, **wide)),
        "compressor0": (ns.compressor, wide), "compressor1": (ns.compressor, wide),
        "combiner0": (ns.conv_combiner, wide), "combiner1": (ns.conv_combiner, wide),
        "xattn2": (ns.xattn_subtract, wide)})
    spec = ns.ModelSpec(nets, name="hybrid_no_ensemble_wide_hp", channels=(7, 7), prefix="moeMerged")
    state = weights.synth_state(spec, seed=29)
    eng = get_engine(engines, "wide7", spec, state, True)
    assert eng.program.fused_read_convolver
    oracle = mo.Oracle(spec, state)
    for seed in (1, 2, 3, 4):
        batch = synth.make_sites(2 + seed, seed=40 + seed, coverage=9 + seed, hybrid_coverage=5 + seed, channels=7, channels1=7)
        assert batch.reads0.shape[2] == 7 and batch.reads1.shape[2] == 7
        logits, _ = eng.forward_batch(batch)
        want, _ = mo.forward_batch(oracle, batch, chunk_sites=len(batch.alleles_per_site))
        np.testing.assert_allclose(logits, want, **LOGIT_TOL)


def _direct_segment_sum(d, slots):
    """Segment sums taken directly (each segment added on its own), instead of the reference's cumulative sum over
    the whole call followed by differences (MixtureOfExpertsAdvanced.py:29-34), whose result for one allele carries
    the rounding of every allele before it in the batch."""
    off = np.concatenate([[0], np.cumsum(np.asarray(slots, dtype=np.int64))])
    return np.stack([d[off[i]:off[i + 1]].sum(axis=0, dtype=np.float32) for i in range(len(off) - 1)]).astype(np.float32)


def test_ragged_extremes(engines, monkeypatch):
    """1-read alleles next to 1000-read alleles, many alleles per site, a single-site batch of one read.  Held to the
    COMMON tolerance against the oracle with direct segment sums; against the literal cumsum-difference form the
    allowance is what that form itself loses next to a 1000-read allele (measured here: literal vs direct oracle)."""
    from oracle import moe_oracle as mo
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=21)
    eng = get_engine(engines, "fresh_single_tech", spec, state, True)
    rng = np.random.default_rng(3)
    rpa = np.array([1, 1000, 1, 3, 2, 1, 1, 257, 5, 1, 1, 1], dtype=np.int32)
    aps = np.array([2, 1, 6, 3], dtype=np.int32)
    reads = rng.integers(0, 255, size=(int(rpa.sum()), 150, 6), dtype=np.uint8)
    reads[0] = 0                                         # an all-zero dummy read
    logits, _ = eng.forward(reads, rpa, aps)
    oracle = mo.Oracle(spec, state)
    literal = oracle.forward((np.transpose(reads, (0, 2, 1)), None), aps, (rpa, None))[:, 0]
    monkeypatch.setattr(mo, "segment_sum", _direct_segment_sum)
    direct = oracle.forward((np.transpose(reads, (0, 2, 1)), None), aps, (rpa, None))[:, 0]
    monkeypatch.undo()
    np.testing.assert_allclose(logits[0], direct, **LOGIT_TOL)
    own_noise = float(np.abs(literal - direct).max())     # the reference arithmetic's distance from order-free sums
    assert np.abs(logits[0] - literal).max() <= own_noise + LOGIT_TOL["atol"] + LOGIT_TOL["rtol"] * np.abs(literal).max()
    print(f"ragged extremes: |gpu - direct| {np.abs(logits[0] - direct).max():.2e}, |literal - direct| {own_noise:.2e}")
    one, _ = eng.forward(reads[:1], np.array([1], np.int32), np.array([1], np.int32))
    want1 = oracle.forward((np.transpose(reads[:1], (0, 2, 1)), None), [1], ([1], None))[:, 0]
    np.testing.assert_allclose(one[0], want1, **LOGIT_TOL)


# ---- randomised stress against the ORACLE: the five BASELINE configurations x weight scales x ragged extremes --------
BASELINE_CONFIGS = [
    ("C1/C2 Illumina 30x", "single_tech", dict(coverage=30)),
    ("C3 PacBio HiFi", "single_tech", dict(coverage=(8, 52), tech="pacbio")),
    ("C4 hybrid no-ensemble", "hybrid_no_ensemble", dict(coverage=30, hybrid_coverage=15)),
    ("C5 haplotagged", "single_tech_hp", dict(coverage=(20, 80), channels=7, tech="pacbio")),
    ("hybrid full (3 experts + meta)", "hybrid_full", dict(coverage=20, hybrid_coverage=10)),
]


def _with_extremes(batch, seed, hybrid, channels):
    """The synthetic batch followed by hand-made extreme sites: a 1-read allele next to a 1000-read allele, a 4-allele
    site whose alleles hold 1 / 257 / 2 / 1 reads, a site whose only support is all-zero dummy reads."""
    rng = np.random.default_rng(seed)
    rpa = np.array([1, 1000, 1, 257, 2, 1, 1, 1], np.int32)
    aps = np.array([2, 4, 2], np.int32)
    def real_reads(n, seed, **kw):
        r = synth.make_sites(n // 20 + 8, seed=seed, coverage=30, **kw).reads0
        assert r.shape[0] >= n
        return r[:n].copy()
    extra = real_reads(int(rpa.sum()), seed, channels=channels)
    extra[0] = 0
    extra[-2:] = 0                                         # the last site: two unsupported alleles
    reads1 = rpa1 = None
    if hybrid:
        rpa1 = np.array([3, 1, 128, 1, 1, 7, 1, 1], np.int32)
        reads1 = real_reads(int(rpa1.sum()), seed + 1, tech="pacbio")
        reads1[-2:] = 0
    ref = synth.make_sites(3, seed=seed + 2, coverage=1).ref_onehot
    cat = lambda a, b: None if a is None else np.concatenate([a, b])        # noqa: E731
    return synth.SiteBatch(cat(batch.reads0, extra), cat(batch.reads_per_allele0, rpa), cat(batch.alleles_per_site, aps),
                           cat(batch.ref_onehot, ref), cat(batch.reads1, reads1), cat(batch.reads_per_allele1, rpa1))


@pytest.mark.parametrize("gain", [0.5, 1.0, 2.5])
@pytest.mark.parametrize("label,cfg,kw", BASELINE_CONFIGS, ids=[c[1] + "-" + str(i) for i, c in enumerate(BASELINE_CONFIGS)])
def test_stress_against_oracle(label, cfg, kw, gain):
    """36 sites + extremes (the round-3 size: more partial workgroups and read-group seams than the 16 of rounds 4-5, whose cases these
    replace -- same generator, same gains)."""
    _stress_against_oracle(label, cfg, kw, gain, 36)


def _stress_against_oracle(label, cfg, kw, gain, n_sites):
    """The fused Winograd engine (F(3,3) on unnormalised 0..254 inputs, sums over up to 1000 reads) against the CPU
    oracle: pair posteriors within the north star's 1e-4, per-allele probabilities within 1e-4, logits within 2e-5 of
    their scale, at three weight scales (logits from O(0.1) to O(1e6))."""
    from hello_amd.engine import Engine
    from oracle import moe_oracle as mo
    spec = ns.build(cfg)
    state = weights.synth_state(spec, seed=77, gain=gain)
    hybrid = "hybrid_coverage" in kw
    batch = _with_extremes(synth.make_sites(n_sites, seed=int(1000 * gain) + len(cfg), **kw), 4000 + int(10 * gain), hybrid,
                           kw.get("channels", 6))
    eng = Engine(spec, state, device=0)
    logits, meta, post = eng.forward_batch(batch, posteriors=True)
    again, _, post2 = eng.forward_batch(batch, posteriors=True)
    assert np.array_equal(logits, again) and np.array_equal(post, post2)            # bit-reproducible
    # the oracle in the reference's DEPLOYMENT form: one site per call (caller_calling.py:872-891).  Its batched form
    # sums by cumulative sum over the whole call and differences (MixtureOfExpertsAdvanced.py:29-34), so a site's
    # result carries the rounding of every site before it: at gain 2.5 (features ~1e6) a site of two identical
    # alleles, whose expert input 2a - s is exactly 0, comes out 0.07 away in probability from the same site scored
    # alone -- the reference's batched form disagreeing with its own per-site form.  The engine sums each segment
    # directly, i.e. agrees with the per-site form, which is what a call through the plug-in surface computes.
    want, want_meta = oracle_per_site(spec, state, batch)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(logits - want).max() <= 2e-5 * scale + 2e-4, (label, gain, float(np.abs(logits - want).max()), scale)
    assert np.abs(sigmoid(logits) - sigmoid(want)).max() < PROB_ATOL
    if want_meta is not None:
        assert np.abs(meta - want_meta).max() < PROB_ATOL
    # pair posteriors as the wrapper computes them (MixtureOfExpertsAdvanced.py:530-589), site by site
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    col, worst = 0, 0.0
    for s in range(batch.n_sites):
        probs = [mo.sigmoid(want[e, aoff[s]:aoff[s + 1]]) for e in range(want.shape[0])]
        if len(probs) == 1:
            probs += [np.zeros_like(probs[0])] * 2
        m = want_meta[s] if want_meta is not None else np.array([1, 0, 0], np.float32)
        rows = mo.posteriors(probs, m)
        n = rows[0].shape[0]
        worst = max(worst, max(float(np.abs(post[r, col:col + n] - rows[r]).max()) for r in range(4)))
        col += n
    assert col == post.shape[1] and worst < PROB_ATOL, (label, gain, worst)
    print(f"stress {label} gain {gain}: |dlogit|/scale {np.abs(logits - want).max() / scale:.2e}, posteriors {worst:.2e}")
    eng.close()


# ---- arithmetic mode bf16x3 (selectable, never the default) ------------------------------------------------------------
def test_bf16x3_mode_is_refused_where_it_does_not_exist():
    from hello_amd.engine import Engine
    for cfg, kw in (("merged_hybrid_250", {}), ("single_tech", dict(winograd=False)), ("single_tech", dict(fused="trunk")),
                    ("single_tech_softplus", {}), ("hybrid_no_ensemble_wide", {})):
        spec = ns.build(cfg)
        with pytest.raises(ValueError, match="bf16x3"):
            Engine(spec, weights.synth_state(spec, seed=1), device=0, arithmetic="bf16x3", **kw)
    with pytest.raises(ValueError, match="arithmetic"):
        Engine(ns.build("single_tech"), weights.synth_state(ns.build("single_tech"), seed=1), device=0, arithmetic="fp16")


_ORACLE_LOGITS = {}


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x3+32"])
@pytest.mark.parametrize("cfg,kw", [("single_tech", dict(coverage=30)), ("hybrid_full", dict(coverage=20, hybrid_coverage=10)),
                                    ("single_tech_hp", dict(coverage=(20, 80), channels=7, tech="pacbio"))])
def test_bf16x3_mode_frames_and_posteriors(cfg, kw, mode):
    """The read convolver's 64-channel trunk on the bf16 matrix cores (3-term splits, fp32 residual stream): its
    per-allele frames against the exact-fp32 engine's (the same kernel up to the strided block) -- a relative deviation at
    the 1e-5 level of the frames' scale, never more than 1e-4 --, bit-reproducible, and the model's posteriors against
    the oracle inside the north star's 1e-4 at the fixtures' weight scale."""
    from hello_amd.engine import Engine
    from oracle import moe_oracle as mo
    spec = ns.build(cfg)
    state = weights.synth_state(spec, seed=21)
    hybrid = "hybrid_coverage" in kw
    batch = _with_extremes(synth.make_sites(40, seed=5 + len(cfg), **kw), 77, hybrid, kw.get("channels", 6))
    exact, split = Engine(spec, state, device=0, arithmetic="fp32"), Engine(spec, state, device=0, arithmetic=mode)
    assert split.program.arithmetic == mode and exact.program.arithmetic == "fp32"
    fused = [[i for i, o in enumerate(e.program.ops) if o.kind == 8] for e in (exact, split)]
    assert len(fused[0]) == len(fused[1]) == (2 if hybrid else 1)
    for i, pair in enumerate(zip(*fused)):       # op numbering differs between programs (the expert front is one op or five)
        frames = []
        for eng, op in zip((exact, split), pair):
            eng.capture_op_output(op)
            eng.forward_batch(batch)
            frames.append(eng.read_op_output().copy())
            eng.capture_op_output(None)
        scale = float(np.abs(frames[0]).max())
        dev = float(np.abs(frames[1] - frames[0]).max()) / scale
        print(f"{mode} {cfg} op {i}: frames max |d| / scale = {dev:.2e} (scale {scale:.3g})")
        assert 0 < dev < 1e-4
    logits, meta, post = split.forward_batch(batch, posteriors=True)
    again, _, post2 = split.forward_batch(batch, posteriors=True)
    assert np.array_equal(logits, again) and np.array_equal(post, post2)
    exact_logits, _, exact_post = exact.forward_batch(batch, posteriors=True)
    if cfg not in _ORACLE_LOGITS:                # the oracle's answer does not depend on the engine's mode: once per model
        _ORACLE_LOGITS[cfg] = oracle_per_site(spec, state, batch)[0]
    want = _ORACLE_LOGITS[cfg]
    d_oracle = float(np.abs(sigmoid(logits) - sigmoid(want)).max())
    d_exact = float(np.abs(post - exact_post).max())
    print(f"{mode} {cfg}: allele probabilities vs oracle {d_oracle:.2e}, posteriors vs the fp32 engine {d_exact:.2e}")
    assert d_oracle < PROB_ATOL and d_exact < PROB_ATOL
    exact.close()
    split.close()


def test_a_site_scored_alone_and_inside_a_large_launch_agree_to_rounding():
    """launches that cannot fill the chip take the small-launch kernel family (other K-summation order), and a
    site's per-allele read sums are cut where the fused kernel's workgroups end, so a site's logits depend on the launch it
    is part of -- at rounding level only: one site per call against the same sites inside a 2 048-site launch."""
    from hello_amd.engine import Engine
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=9)
    eng = Engine(spec, state, device=0, arithmetic="fp32")
    batch = synth.make_sites(2048, seed=12, coverage=30)
    big, _ = eng.forward_batch(batch)
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    worst = 0.0
    for s in (0, 1, 511, 1024, 2046, 2047):
        alone, _ = eng.forward_batch(batch.site_slice(s, s + 1))
        ref = big[:, aoff[s]:aoff[s + 1]]
        worst = max(worst, float(np.abs(alone - ref).max() / max(1.0, np.abs(ref).max())))
    print(f"one site per call vs inside a 2 048-site launch: max |d logit| / scale = {worst:.2e}")
    assert worst < 2e-6
    eng.close()


@pytest.mark.parametrize("cfg", ["single_tech", "hybrid_full"])
def test_sites_with_many_alleles(cfg):
    """Sites of 9, 12 and 17 alleles between ordinary ones: the posteriors kernel's literal path (more than 8 alleles: 45 / 78 / 153
    pairs per site), site sums over more alleles than a workgroup of the fused expert front holds (8 items: a 12-allele site spans
    two of them, a 17-allele site three), the compressor's partly filled last workgroup.  Logits, meta and all four posterior rows
    against the oracle in the reference's deployment form (one site per call)."""
    from hello_amd.engine import Engine
    from oracle import moe_oracle as mo
    spec = ns.build(cfg)
    state = weights.synth_state(spec, seed=31)
    hybrid = spec.hybrid_inputs
    rng = np.random.default_rng(17)
    aps = np.array([2, 9, 1, 12, 3, 17, 2], np.int32)
    A = int(aps.sum())
    rpa0 = rng.integers(1, 5, size=A).astype(np.int32)
    rpa1 = rng.integers(1, 4, size=A).astype(np.int32) if hybrid else None
    pool = synth.make_sites(40, seed=5, coverage=30, **(dict(hybrid_coverage=15) if hybrid else {}))
    reads0 = pool.reads0[:int(rpa0.sum())].copy()
    reads1 = pool.reads1[:int(rpa1.sum())].copy() if hybrid else None
    reads0[int(rpa0[:5].sum())] = 0                                      # an unsupported allele's dummy read inside the 9-allele site
    batch = synth.SiteBatch(reads0, rpa0, aps, pool.ref_onehot[:aps.shape[0]].copy(), reads1, rpa1)
    eng = Engine(spec, state, device=0)
    logits, meta, post = eng.forward_batch(batch, posteriors=True)
    want, want_meta = oracle_per_site(spec, state, batch)
    np.testing.assert_allclose(logits, want, **LOGIT_TOL)
    if want_meta is not None:
        assert np.abs(meta - want_meta).max() < PROB_ATOL
    aoff = np.concatenate([[0], np.cumsum(aps)])
    col = 0
    for s in range(aps.shape[0]):
        probs = [mo.sigmoid(want[e, aoff[s]:aoff[s + 1]]) for e in range(want.shape[0])]
        if len(probs) == 1:
            probs += [np.zeros_like(probs[0])] * 2
        m = want_meta[s] if want_meta is not None else np.array([1, 0, 0], np.float32)
        rows = mo.posteriors(probs, m)
        n = rows[0].shape[0]
        assert n == aps[s] * (aps[s] + 1) // 2
        for r in range(4):
            assert np.abs(post[r, col:col + n] - rows[r]).max() < PROB_ATOL, (s, r)
        col += n
    assert col == post.shape[1] == 45 + 78 + 153 + 3 + 1 + 6 + 3
    # the same sites one per call through the small-launch kernels
    for s in (1, 3, 5):
        one, _, p1 = eng.forward_batch(batch.site_slice(s, s + 1), posteriors=True)
        np.testing.assert_allclose(one, want[:, aoff[s]:aoff[s + 1]], **LOGIT_TOL)
    eng.close()


def test_stamped_read_convolver_gives_the_same_bits_and_a_sane_timeline():
    """hello_engine_debug_stamps (DESIGN 3.1's measured cost model rests on it): the stamped instantiation of the fused read convolver
    computes the same bits as the product kernel, at two workgroups per CU and at one; every (workgroup, wave, group) record is
    monotone -- start <= arrival <= release at each of the 20 barriers -- and the layout says how the launches covered the reads.
    Models outside the canonical fp32 Winograd schedule refuse the diagnostic instead of recording nonsense."""
    from hello_amd.engine import Engine
    spec = ns.build("single_tech")
    state = weights.synth_state(spec, seed=3)
    batch = synth.make_sites(700, seed=17, coverage=30)              # a bulk launch of several groups per workgroup + a remainder launch
    eng = Engine(spec, state, device=0, arithmetic="fp32")
    want, _, want_post = eng.forward_batch(batch, posteriors=True)
    for mode in (1, 3):
        eng.record_stamps(mode)
        got, _, got_post = eng.forward_batch(batch, posteriors=True)
        assert np.array_equal(got, want) and np.array_equal(got_post, want_post)
        stamps, bulk = eng.read_stamps()
        wgs, waves, groups, slots = stamps.shape
        assert waves == 4 and slots >= 45 and 0 < bulk <= wgs and groups >= 1
        n_groups = (batch.reads0.shape[0] + 3) // 4
        s = stamps.astype(np.int64)
        live = s[..., 0] > 0
        assert int(live.sum()) == 4 * n_groups                        # every group of 4 reads left one record per wave
        assert live[bulk:, :, 1:].sum() == 0                          # the remainder launch's workgroups walk one group
        seq = np.concatenate([s[..., 0:1], s[..., 1:41]], axis=-1)[live]
        assert (np.diff(seq, axis=-1) >= 0).all()                     # start <= arrive 0 <= release 0 <= arrive 1 <= ...
        assert (s[..., 43][live] > s[..., 42][live]).all()            # the 100 MHz clock moved inside every group
        total = (s[..., 40] - s[..., 0])[live]
        assert 5e4 < np.median(total) < 2e6                           # a group is ~1.5e5 (one wave per SIMD) to ~2.6e5 cycles
    eng.record_stamps(0)
    again, _, _ = eng.forward_batch(batch, posteriors=True)
    assert np.array_equal(again, want)
    with pytest.raises(RuntimeError, match="mode"):
        eng.record_stamps(2)
    eng.close()
    wide = ns.build("hybrid_no_ensemble", w=2)
    eng = Engine(wide, weights.synth_state(wide, seed=3), device=0)
    eng.record_stamps(1)
    with pytest.raises(RuntimeError, match="stamps"):
        eng.forward_batch(synth.make_sites(5, seed=2, coverage=10, hybrid_coverage=6))
    eng.close()
