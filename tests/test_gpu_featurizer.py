"""GPU: the HIP featurizer (hello_engine_featurize) against the oracle restatement of
computeFeaturesColoredSimple, bit for bit, and feeding the scoring engine without leaving the device."""
import numpy as np
import pytest

from hello_amd import netspec as ns, weights
from hello_amd.featurizer import AlignedRead, SiteReads, featurize
from oracle import featurizer_oracle as fo
from tests.test_featurizer import EXPECTED, REFERENCE, reference_generated_sites, reference_test_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from hello_amd.engine import Engine
    spec = ns.build("single_tech")
    eng = Engine(spec, weights.synth_state(spec, seed=2))
    yield eng
    eng.close()


def to_site(reads_by_allele, reference, window_start, a0, a1):
    return SiteReads(reference, window_start, a0, a1,
                     [(name, [AlignedRead(r.bases, r.quals, r.cigar, r.ref_start, r.mapq, r.orientation, r.hp) for r in reads])
                      for name, reads in reads_by_allele])


def test_reference_unit_test_case(engine):
    for tagged in (False, True):
        reads = reference_test_reads(tagged)
        site = to_site([("x", reads)], REFERENCE, 0, 10, 14)
        out, rpa, aps = featurize(engine, [site], feature_length=10, include_hp=tagged)
        np.testing.assert_array_equal(out[:, :, :6], EXPECTED)
        np.testing.assert_array_equal(out, fo.features_for_reads(reads, REFERENCE, 0, 10, 14, 10, tagged))
        assert rpa.tolist() == [3] and aps.tolist() == [1]


def random_read(rng, window_start, ref_len, span):
    """A read with a random CIGAR somewhere around the allele span (may start before / end after the window)."""
    ops, n_bases, ref_len_used = [], 0, 0
    if rng.random() < 0.3:
        k = int(rng.integers(1, 6)); ops.append((fo.BAM_CSOFT_CLIP, k)); n_bases += k
    for _ in range(int(rng.integers(1, 7))):
        k = int(rng.integers(1, 60)); ops.append((int(rng.choice([fo.BAM_CMATCH, fo.BAM_CEQUAL, fo.BAM_CDIFF])), k))
        n_bases += k; ref_len_used += k
        u = rng.random()
        if u < 0.25:
            k = int(rng.integers(1, 12)); ops.append((fo.BAM_CINS, k)); n_bases += k
        elif u < 0.5:
            k = int(rng.integers(1, 20)); ops.append((fo.BAM_CDEL, k)); ref_len_used += k
        elif u < 0.55:
            k = int(rng.integers(1, 30)); ops.append((fo.BAM_CREF_SKIP, k)); ref_len_used += k
    if rng.random() < 0.2:
        k = int(rng.integers(1, 6)); ops.append((fo.BAM_CSOFT_CLIP, k)); n_bases += k
    if rng.random() < 0.1:                                   # an insertion / deletion as the very first operation
        ops.insert(0, (int(rng.choice([fo.BAM_CINS, fo.BAM_CDEL])), 3))
        if ops[0][0] == fo.BAM_CINS:
            n_bases += 3
        else:
            ref_len_used += 3
    lo = max(window_start + 1, span[0] - ref_len_used)
    start = int(rng.integers(lo, max(lo + 1, span[1])))
    start = min(start, window_start + ref_len - ref_len_used - 1)
    bases = "".join(rng.choice(list("ACGTN"), size=n_bases, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
    quals = rng.integers(0, 60, size=n_bases).tolist()
    return fo.Read(bases, quals, ops, start, mapq=int(rng.integers(0, 90)), orientation=int(rng.choice([-1, 1])),
                   hp=int(rng.integers(0, 3)))


@pytest.mark.parametrize("include_hp,length", [(False, 150), (True, 150), (False, 37)])
def test_random_reads_match_oracle_bit_for_bit(engine, include_hp, length):
    rng = np.random.default_rng(11 + length + include_hp)
    sites, want = [], []
    for s in range(40):
        window_start = int(rng.integers(1000, 5000))
        ref_len = 1400
        reference = "".join(rng.choice(list("ACGT"), size=ref_len))
        a0 = window_start + int(rng.integers(500, 800))
        a1 = a0 + int(rng.integers(1, 12))
        alleles = []
        for k in range(int(rng.integers(1, 4))):
            n = int(rng.choice([0, 1, 2, 5, 9]))
            reads = [random_read(rng, window_start, ref_len, (a0, a1)) for _ in range(n)]
            alleles.append((f"al{k}", reads))
            want.append(fo.features_for_reads(reads, reference, window_start, a0, a1, length, include_hp))
        sites.append(to_site(alleles, reference, window_start, a0, a1))
    out, rpa, aps = featurize(engine, sites, feature_length=length, include_hp=include_hp)
    want = np.concatenate(want, axis=0)
    assert out.shape == want.shape and int(rpa.sum()) == want.shape[0] and int(aps.sum()) == rpa.shape[0]
    np.testing.assert_array_equal(out, want)
    assert (want != 0).any()


def test_features_stay_on_the_device_into_the_scoring_engine(engine):
    import torch
    rng = np.random.default_rng(3)
    sites = []
    for s in range(6):
        reference = "".join(rng.choice(list("ACGT"), size=600))
        alleles = [(f"a{k}", [random_read(rng, 100, 600, (380, 384)) for _ in range(int(rng.integers(0, 7)))])
                   for k in range(2)]
        sites.append(to_site(alleles, reference, 100, 380, 384))
    dev, rpa, aps = featurize(engine, sites, device_output=True)
    host, _, _ = featurize(engine, sites)
    assert dev.is_cuda and torch.equal(dev.cpu(), torch.from_numpy(host))
    a, _ = engine.forward(dev, rpa, aps)
    b, _ = engine.forward(host, rpa, aps)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a.cpu().numpy(), b)


def test_reference_generated_fixtures_bit_exact(engine):
    """The HIP featurizer against what the REFERENCE's own encoder produced (featurizer_reference.npz): its two
    unit-test cases and 336 random reads, 6 and 7 channels, windows of 150 / 33 / 10 -- one launch per (length,
    channels) group, every site of the group in the same launch."""
    sites = reference_generated_sites()
    groups = {}
    for site in sites:
        groups.setdefault((site[3], site[4]), []).append(site)
    for (length, tagged), members in groups.items():
        packed = [to_site([("x", reads)], reference, 0, a0, a1) for reference, a0, a1, _, _, reads, _ in members]
        out, rpa, aps = featurize(engine, packed, feature_length=length, include_hp=tagged)
        np.testing.assert_array_equal(out, np.concatenate([m[6] for m in members]))
        assert rpa.tolist() == [len(m[5]) for m in members]
