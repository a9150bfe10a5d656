"""GPU: the per-site plug-in surface (hello_amd.wrapper / hello_amd.loader) against what the reference's
MoEMergedWrapperAdvanced returned for the same sites (golden vectors), through the C ABI."""
import os

import numpy as np
import pytest

from hello_amd import loader, synth
from tests.util import GOLDEN, load_fixture

pytestmark = pytest.mark.gpu
PROB = dict(rtol=2e-4, atol=1e-4)


def site_dicts(batch, as_float=True):
    import torch
    names = synth.allele_names(batch)
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    r0 = np.concatenate([[0], np.cumsum(batch.reads_per_allele0)])
    r1 = None if batch.reads1 is None else np.concatenate([[0], np.cumsum(batch.reads_per_allele1)])
    out = []
    for s in range(batch.n_sites):
        fd = {}
        for j, a in enumerate(range(aoff[s], aoff[s + 1])):
            # exactly what caller_calling.scoreSite builds: torch.Tensor(uint8 ndarray) -> float32
            t0 = torch.Tensor(batch.reads0[r0[a]:r0[a + 1]])
            t1 = None if r1 is None else torch.Tensor(batch.reads1[r1[a]:r1[a + 1]])
            fd[names[s][j]] = (t0, t1)
        out.append((fd, torch.from_numpy(batch.ref_onehot[s:s + 1]).float()))
    return out


@pytest.mark.parametrize("name", ["single_tech_batched", "single_tech_hp", "hybrid_no_ensemble", "hybrid_full",
                                  "hybrid_ensemble2", "merged_single", "merged_hybrid"])
def test_per_site_call_matches_reference_wrapper(name):
    import torch
    from hello_amd.wrapper import ScoringNetwork
    spec, state, batch, exp = load_fixture(name)
    net = ScoringNetwork(spec, state)
    assert net.eval() is net
    sites = site_dicts(batch)
    # default: dict of pair -> 0-dim tensor
    d = net(*sites[0])
    assert list("|".join(k) for k in d) == list(exp["site0_pairs"])
    assert all(isinstance(v, torch.Tensor) and v.dim() == 0 for v in d.values())
    net.providePredictions = True
    for s, (fd, seg) in enumerate(sites):
        mix, e0, e1, e2, meta = net(fd, seg)
        for got, key in ((mix, "mix"), (e0, "e0"), (e1, "e1"), (e2, "e2")):
            np.testing.assert_allclose(np.array([float(v) for v in got.values()]), exp[f"site{s}_{key}"], **PROB)
        np.testing.assert_allclose(meta.numpy(), exp[f"site{s}_meta"], **PROB)
    # throughput form: all sites in one launch, same answers, same order
    many = net.score_sites(sites)
    for s, res in enumerate(many):
        np.testing.assert_allclose(np.array([float(v) for v in res[0].values()]), exp[f"site{s}_mix"], **PROB)
    net.close()


def test_batched_operator_surface_matches_reference_layout():
    """network.moeMerged(tensors [sumR, C, L], alleles/site, reads/allele, ref) -> logits [sumA, 1]."""
    import torch
    from hello_amd.wrapper import ScoringNetwork
    spec, state, batch, exp = load_fixture("hybrid_full")
    net = ScoringNetwork(spec, state)
    t0 = torch.from_numpy(np.ascontiguousarray(np.transpose(batch.reads0, (0, 2, 1))))
    t1 = torch.from_numpy(np.ascontiguousarray(np.transpose(batch.reads1, (0, 2, 1))))
    experts, meta = net.moeMerged((t0, t1), batch.alleles_per_site.tolist(),
                                  (batch.reads_per_allele0.tolist(), batch.reads_per_allele1.tolist()),
                                  torch.from_numpy(batch.ref_onehot).float(), "ignored extra positional")
    assert len(experts) == 3 and experts[0].shape == (batch.n_alleles, 1)
    np.testing.assert_allclose(np.stack([e[:, 0].numpy() for e in experts]), exp["logits"], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(meta.numpy(), exp["meta"], **PROB)
    net.close()


def test_loader_runs_a_real_reference_pickle():
    net = loader.load(os.path.join(GOLDEN, "mini_reference.wrapper.dnn"))
    net.eval()
    net.providePredictions = True
    z = np.load(os.path.join(GOLDEN, "mini_reference.npz"))
    batch = synth.SiteBatch(z["reads0"], z["reads_per_allele0"], z["alleles_per_site"], z["ref_onehot"])
    logits, _ = net.engine.forward_batch(batch)
    np.testing.assert_allclose(logits, z["exp_logits"], rtol=2e-5, atol=2e-5)
    for s, (fd, seg) in enumerate(site_dicts(batch)):
        mix, e0, _, _, meta = net(fd, seg)
        np.testing.assert_allclose(np.array([float(v) for v in mix.values()]), z[f"exp_site{s}_mix"], **PROB)
        assert meta.tolist() == [1.0, 0.0, 0.0]
    net.close()


def test_loader_runs_a_transfer_learning_pickle():
    net = loader.load(os.path.join(GOLDEN, "mini_addendum.wrapper.dnn"))
    z = np.load(os.path.join(GOLDEN, "mini_addendum.npz"))
    batch = synth.SiteBatch(z["reads0"], z["reads_per_allele0"], z["alleles_per_site"], z["ref_onehot"])
    logits, _ = net.engine.forward_batch(batch)
    np.testing.assert_allclose(logits, z["exp_logits"], rtol=2e-5, atol=2e-5)
    for s, (fd, seg) in enumerate(site_dicts(batch)):
        d = net(fd, seg)
        np.testing.assert_allclose(np.array([float(v) for v in d.values()]), z[f"exp_site{s}_mix"], **PROB)
    net.close()


def test_loader_runs_a_merged_family_pickle():
    """MoEMergedAdvanced (hybrid, additive, combiners, BatchNorm meta) straight from a reference pickle."""
    net = loader.load(os.path.join(GOLDEN, "mini_merged.wrapper.dnn"))
    net.providePredictions = True
    z = np.load(os.path.join(GOLDEN, "mini_merged.npz"))
    batch = synth.SiteBatch(z["reads0"], z["reads_per_allele0"], z["alleles_per_site"], z["ref_onehot"],
                            z["reads1"], z["reads_per_allele1"])
    logits, meta = net.engine.forward_batch(batch)
    np.testing.assert_allclose(logits, z["exp_logits"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(meta, z["exp_meta"], rtol=1e-5, atol=2e-6)
    for s, (fd, seg) in enumerate(site_dicts(batch)):
        mix, e0, e1, e2, m = net(fd, seg)
        for got, key in ((mix, "mix"), (e0, "e0"), (e1, "e1"), (e2, "e2")):
            np.testing.assert_allclose(np.array([float(v) for v in got.values()]), z[f"exp_site{s}_{key}"], **PROB)
        np.testing.assert_allclose(m.numpy(), z[f"exp_site{s}_meta"], **PROB)
    net.close()


def test_loader_runs_the_hybrid_compressor_branch_pickle():
    """MixtureOfExpertsAdvanced.py:181-192 (``compressor2`` + ``xattn2``, meta on the hybrid compressor's site-level
    output) from a reference pickle: logits, meta and the wrapper's five outputs against what the reference returned."""
    net = loader.load(os.path.join(GOLDEN, "mini_compressor2.wrapper.dnn"))
    net.providePredictions = True
    z = np.load(os.path.join(GOLDEN, "mini_compressor2.npz"))
    batch = synth.SiteBatch(z["reads0"], z["reads_per_allele0"], z["alleles_per_site"], z["ref_onehot"], z["reads1"],
                            z["reads_per_allele1"])
    logits, meta = net.engine.forward_batch(batch)
    np.testing.assert_allclose(logits, z["exp_logits"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(meta, z["exp_meta"], rtol=1e-5, atol=2e-6)
    for s, (fd, seg) in enumerate(site_dicts(batch)):
        mix, e0, e1, e2, m = net(fd, seg)
        for got, key in ((mix, "mix"), (e0, "e0"), (e1, "e1"), (e2, "e2")):
            np.testing.assert_allclose(np.array([float(v) for v in got.values()]), z[f"exp_site{s}_{key}"], **PROB)
        np.testing.assert_allclose(m.numpy(), z[f"exp_site{s}_meta"], **PROB)
    net.close()


@pytest.mark.parametrize("name", ["mini_merged_concat", "mini_merged_sepmeta"])
def test_loader_runs_the_other_merged_family_variants(name):
    """mini_merged_concat: the class default useAdditive=False (expert input cat(allele, rest of site));
    mini_merged_sepmeta: useSeparateMeta (the meta-expert's own read convolvers, per-site read sums), a BatchNorm
    eps other than 1e-5, a BatchNorm without affine parameters, a convolution without bias -- all straight from the
    reference's pickles, against what the reference returned."""
    net = loader.load(os.path.join(GOLDEN, name + ".wrapper.dnn"))
    net.providePredictions = True
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    hybrid = "reads1" in z.files
    batch = synth.SiteBatch(z["reads0"], z["reads_per_allele0"], z["alleles_per_site"], z["ref_onehot"],
                            z["reads1"] if hybrid else None, z["reads_per_allele1"] if hybrid else None)
    logits, meta = net.engine.forward_batch(batch)
    np.testing.assert_allclose(logits, z["exp_logits"], rtol=2e-5, atol=2e-5)
    if hybrid:
        np.testing.assert_allclose(meta, z["exp_meta"], rtol=1e-5, atol=2e-6)
    for s, (fd, seg) in enumerate(site_dicts(batch)):
        mix, e0, e1, e2, m = net(fd, seg)
        for got, key in ((mix, "mix"), (e0, "e0"), (e1, "e1"), (e2, "e2")):
            np.testing.assert_allclose(np.array([float(v) for v in got.values()]), z[f"exp_site{s}_{key}"], **PROB)
        np.testing.assert_allclose(m.numpy(), z[f"exp_site{s}_meta"], **PROB)
    net.close()


def test_site_batcher_preserves_order():
    from hello_amd.wrapper import ScoringNetwork, SiteBatcher
    spec, state, batch, exp = load_fixture("single_tech_batched")
    net = ScoringNetwork(spec, state)
    batcher = SiteBatcher(net, max_sites=4)
    got = []
    for s, (fd, seg) in enumerate(site_dicts(batch)):
        got += batcher.submit(fd, seg, tag=s)
    got += batcher.flush()
    assert [t for t, _ in got] == list(range(batch.n_sites))
    for s, res in got:
        np.testing.assert_allclose(np.array([float(v) for v in res.values()]), exp[f"site{s}_mix"], **PROB)
    net.close()


def test_non_integer_pileups_are_rejected():
    import torch
    from hello_amd.wrapper import ScoringNetwork
    spec, state, batch, _ = load_fixture("single_tech_batched")
    net = ScoringNetwork(spec, state)
    fd, seg = site_dicts(batch)[0]
    k = next(iter(fd))
    fd[k] = (fd[k][0] + 0.5, None)
    with pytest.raises(ValueError, match="integers"):
        net(fd, seg)
    net.close()


def test_the_ctypes_stub_of_integration_md_runs_as_written():
    """INTEGRATION.md section 3 shows the binding a maintainer would write against include/hello_mi355x.h: the code block
    is executed verbatim (create, one host-path forward, destroy) and its logits must equal the maintained binding's."""
    import os
    import re
    from hello_amd import compiler, netspec as ns, synth, weights as wts
    from hello_amd.engine import Engine, model_desc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    section = text[text.index("## 3. Binding the C ABI directly"):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    spec = ns.build("single_tech")
    state = wts.synth_state(spec, seed=4)
    batch = synth.make_sites(12, seed=8, coverage=10)
    program = compiler.compile_model(spec, state)
    desc, keep = model_desc(program)
    names = dict(desc=desc, weights=np.ascontiguousarray(program.weights, np.float32), n_experts=program.n_experts,
                 reads0=np.ascontiguousarray(batch.reads0), reads_per_allele0=np.ascontiguousarray(batch.reads_per_allele0, np.int32),
                 alleles_per_site=np.ascontiguousarray(batch.alleles_per_site, np.int32), S=batch.n_sites, A=batch.n_alleles,
                 R0=batch.reads0.shape[0])
    cwd = os.getcwd()
    os.chdir(root)                                    # the stub names the library relative to the repository root
    try:
        exec(compile(code, "INTEGRATION.md section 3", "exec"), names)
    finally:
        os.chdir(cwd)
    eng = Engine(spec, state, device=0, program=program)
    want, _ = eng.forward_batch(batch)
    eng.close()
    assert names["logits"].shape == want.shape and np.array_equal(names["logits"], want)


@pytest.mark.parametrize("cfg,fixture", [("single_tech", "single_tech_batched"), ("hybrid_full", "hybrid_full")])
def test_canonical_size_reference_pickle_reaches_the_fused_kernels(cfg, fixture, tmp_path, suite_arithmetic):
    """a canonical-architecture pickle written by the reference's torch.save (parameters zeroed in the
    committed file: the structure is what it pins) loads without the reference's source, and -- with the fixture's seeded weights
    injected -- the engine built from the PICKLE's spec runs the fused read convolver / compressor / expert front and reproduces
    what the reference returned for the configuration-built model: logits, meta and the wrapper's per-site 5-tuple."""
    from hello_amd import compiler
    from hello_amd.wrapper import ScoringNetwork
    from tests.util import canonical_pickle
    spec, zeroed = loader.load_spec(canonical_pickle(cfg, tmp_path))
    _, state, batch, exp = load_fixture(fixture)
    assert set(zeroed) == set(state)
    net = ScoringNetwork(spec, state)
    prog = net.engine.program
    kinds = [o.kind for o in prog.ops]
    assert prog.fused_read_convolver and prog.fused_compressor and compiler.OP_XATTN_FRONT in kinds and prog.arithmetic == suite_arithmetic
    logits, meta = net.engine.forward_batch(batch)
    np.testing.assert_allclose(logits, exp["logits"], rtol=2e-5, atol=2e-4)
    if "meta" in exp:
        np.testing.assert_allclose(meta, exp["meta"], **PROB)
    net.providePredictions = True
    for s, (fd, seg) in enumerate(site_dicts(batch)):
        if f"site{s}_mix" not in exp:
            break
        mix, e0, e1, e2, m = net(fd, seg)
        np.testing.assert_allclose(np.array([float(v) for v in mix.values()]), exp[f"site{s}_mix"], **PROB)
        np.testing.assert_allclose(m.numpy(), exp[f"site{s}_meta"], **PROB)
    net.close()
