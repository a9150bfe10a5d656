"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz, produced by
tests/golden/make_fixtures.py from /root/reference in the build container)."""
import os

import numpy as np
import pytest

from hello_amd import synth
from oracle import moe_oracle as mo
from tests.util import FIXTURES, GOLDEN, load_fixture

# reference self-noise between its own per-site and batched calls is <= 4e-6 on logits
LOGIT_TOL = dict(rtol=2e-5, atol=2e-5)
PROB_TOL = dict(rtol=1e-5, atol=1e-6)


def _batched(oracle, batch):
    t0 = np.transpose(batch.reads0, (0, 2, 1))
    t1 = None if batch.reads1 is None else np.transpose(batch.reads1, (0, 2, 1))
    out = oracle.forward((t0, t1), batch.alleles_per_site,
                         (batch.reads_per_allele0, batch.reads_per_allele1), batch.ref_onehot)
    if isinstance(out, tuple):
        return np.stack([e[:, 0] for e in out[0]]), out[1]
    return out[:, 0][None, :], None


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("backend", ["numpy", "torch"])
def test_batched_logits_match_reference(name, backend):
    spec, state, batch, exp = load_fixture(name)
    oracle = mo.Oracle(spec, state, backend=backend)
    logits, meta = _batched(oracle, batch)
    np.testing.assert_allclose(logits, exp["logits"], **LOGIT_TOL)
    if "meta" in exp:
        np.testing.assert_allclose(meta, exp["meta"], **PROB_TOL)
    if "frames0" in exp:
        scale = np.abs(exp["frames0"]).max()
        np.testing.assert_allclose(oracle.last["frames0"], exp["frames0"], rtol=1e-5, atol=1e-6 * scale)


def test_read_convolver_frames_match_reference():
    """Kernel-level pins (tests/golden/frames.npz): reduceSlots(read_convolver(x)) of both technologies, as the
    reference computed them on the committed inputs -- 7 channels, 90-128 reads per site, 250 bp, Softplus, the
    transfer-learning blocks, BatchNorm."""
    z = np.load(os.path.join(GOLDEN, "frames.npz"))
    cases = sorted({k.rsplit("_frames", 1)[0] for k in z.files})
    assert len(cases) >= 8
    for name in cases:
        spec, state, batch, _ = load_fixture(name)
        oracle = mo.Oracle(spec, state)
        _batched(oracle, batch)
        for tech in (0, 1):
            key = f"{name}_frames{tech}"
            if key in z.files:
                want = z[key]
                np.testing.assert_allclose(oracle.last[f"frames{tech}"], want, rtol=1e-5, atol=1e-6 * np.abs(want).max())


@pytest.mark.parametrize("name", [f for f in FIXTURES if f not in ("single_tech_bn", "single_tech_deep")])
def test_wrapper_posteriors_match_reference(name):
    spec, state, batch, exp = load_fixture(name)
    wrapper = mo.WrapperOracle(spec, state)
    names = synth.allele_names(batch)
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    r0 = np.concatenate([[0], np.cumsum(batch.reads_per_allele0)])
    r1 = None if batch.reads1 is None else np.concatenate([[0], np.cumsum(batch.reads_per_allele1)])
    for s in range(batch.n_sites):
        fd = {}
        for j, a in enumerate(range(aoff[s], aoff[s + 1])):
            second = None if r1 is None else batch.reads1[r1[a]:r1[a + 1]].astype(np.float32)
            fd[names[s][j]] = (batch.reads0[r0[a]:r0[a + 1]].astype(np.float32), second)
        mix, e0, e1, e2, meta = wrapper(fd, batch.ref_onehot[s:s + 1].astype(np.float32))
        keys = ["|".join(k) for k in mix]
        assert keys == list(exp[f"site{s}_pairs"])          # same pairs, same first-seen order
        for got, want in ((mix, "mix"), (e0, "e0"), (e1, "e1"), (e2, "e2")):
            np.testing.assert_allclose(np.array(list(got.values())), exp[f"site{s}_{want}"], **PROB_TOL)
        np.testing.assert_allclose(meta, exp[f"site{s}_meta"], **PROB_TOL)


def test_segment_sum_matches_direct_sums():
    rng = np.random.default_rng(0)
    d = rng.normal(size=(17, 3, 5)).astype(np.float32)
    slots = [1, 4, 2, 7, 3]
    got = mo.segment_sum(d, slots)
    off = np.concatenate([[0], np.cumsum(slots)])
    want = np.stack([d[off[i]:off[i + 1]].sum(axis=0) for i in range(len(slots))])
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_pair_order_is_first_seen_product_order():
    assert mo.pair_order(3) == [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    assert mo.pair_order(1) == [(0, 0)]


def test_work_per_unit_matches_survey():
    """SURVEY.md 8d / BASELINE.md 3: MACs are shape-determined."""
    from hello_amd import netspec as ns
    spec = ns.build("hybrid_full")
    assert ns.macs(spec.nets["read_convolver0"], 150) == 5_076_096
    assert ns.macs(ns.build("single_tech_hp").nets["read_convolver0"], 150) == 5_083_200
    assert ns.macs(spec.nets["compressor0"], 36) == 5_160_960
    assert ns.macs(spec.nets["xattn0"], 18) == 10_322_176
    assert ns.macs(spec.nets["combiner0"], 18) == 8_257_536
    assert ns.macs(spec.nets["meta"], 18) == 10_322_688
    assert ns.macs(ns.build("hybrid_ensemble2").nets["meta"], 150) == 6_008_288
