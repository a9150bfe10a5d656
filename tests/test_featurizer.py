"""Pileup-tensor producer: the oracle restatement of computeFeaturesColoredSimple pinned by
known answers worked out by hand on the inputs of the reference's own unit test
(reference python/test_aligner.py:279-384: one deletion, one insertion, one mismatch read over
"ACGATACCGTACGGATCGGATCGT", feature length 10, allele span [10, 14))."""
import os

import numpy as np
import pytest

from oracle import featurizer_oracle as fo
from tests.util import GOLDEN

REFERENCE = "ACGATACCGTACGGATCGGATCGT"
M, I, D = fo.BAM_CMATCH, fo.BAM_CINS, fo.BAM_CDEL


def reference_test_reads(tagged):
    hp = (1, 0, 2) if tagged else (0, 0, 0)
    return [
        fo.Read("TAATCG", [26] * 6, [(M, 2), (D, 3), (M, 4)], 9, mapq=30, orientation=-1, hp=hp[0]),
        fo.Read("TAACGGATCG", [30] * 10, [(M, 2), (I, 1), (M, 7)], 9, mapq=44, orientation=1, hp=hp[1]),
        fo.Read("TGCGGATCG", [15] * 9, [(M, 9)], 9, mapq=75, orientation=1, hp=hp[2]),
    ]


Z = [0, 0, 0, 0, 0, 0]
EXPECTED = np.array([
    # read 0: TA, 3-base deletion (the base before it repainted as a gap carrying the previous quality), ATCG
    [Z, Z, [100, 100, 165, 127, 240, 70], [0, 250, 165, 127, 240, 240], [0, 30, 0, 127, 240, 240],
     [0, 180, 0, 127, 240, 240], [0, 180, 0, 127, 240, 240], [250, 250, 165, 127, 240, 70],
     [100, 100, 165, 127, 240, 70], [30, 30, 165, 127, 240, 70]],
    # read 1: TA, 1-base insertion painted as a gap on the base before it, CGGATCG
    [Z, Z, [100, 100, 190, 186, 70, 70], [0, 250, 190, 186, 70, 240], [30, 30, 190, 186, 70, 240],
     [180, 180, 190, 186, 70, 240], [180, 180, 190, 186, 70, 240], [250, 250, 190, 186, 70, 70],
     [100, 100, 190, 186, 70, 70], [30, 30, 190, 186, 70, 70]],
    # read 2: one mismatch (G over A), mapping quality capped at 60
    [Z, Z, [100, 100, 95, 254, 70, 70], [180, 250, 95, 254, 70, 240], [30, 30, 95, 254, 70, 240],
     [180, 180, 95, 254, 70, 240], [180, 180, 95, 254, 70, 240], [250, 250, 95, 254, 70, 70],
     [100, 100, 95, 254, 70, 70], [30, 30, 95, 254, 70, 70]],
], dtype=np.uint8)


def test_oracle_reproduces_hand_worked_reference_case():
    got = fo.features_for_reads(reference_test_reads(False), REFERENCE, 0, 10, 14, 10, include_hp=False)
    np.testing.assert_array_equal(got, EXPECTED)


def test_haplotag_channel():
    got = fo.features_for_reads(reference_test_reads(True), REFERENCE, 0, 10, 14, 10, include_hp=True)
    np.testing.assert_array_equal(got[:, :, :6], EXPECTED)
    written = EXPECTED[:, :, 3] != 0                      # the tag is painted wherever the read paints anything
    for n, colour in enumerate((120, 0, 240)):
        np.testing.assert_array_equal(got[n, :, 6], np.where(written[n], colour, 0))


def test_unsupported_allele_is_one_zero_read():
    got = fo.features_for_reads([], REFERENCE, 0, 10, 14, 10, include_hp=False)
    assert got.shape == (1, 10, 6) and not got.any()


def test_integer_quality_colour_equals_the_double_expression():
    """The HIP kernel computes (254*min(q,cap))/cap in integers; the reference int(254*(1.0*min(q,cap)/cap))."""
    for cap in (40, 60):
        for q in range(256):
            assert (254 * min(q, cap)) // cap == fo.quality_color(q, cap)


def test_colour_alphabet_matches_the_synthetic_generator():
    from hello_amd import synth
    assert [fo.base_color(b) for b in "ACGT"] == synth.BASE_CODE.tolist()
    assert fo.base_color("*") == 0 and fo.base_color("N") == 0


def reference_generated_sites():
    """tests/golden/featurizer_reference.npz: reads + the output of the REFERENCE's own Python encoder
    (python/test_aligner.py:15-180, executed in the build container by tests/golden/make_fixtures.py) for its two
    unit-test cases and 336 random reads.  -> [(reference, a0, a1, length, tagged, [fo.Read], expected uint8)]"""
    z = np.load(os.path.join(GOLDEN, "featurizer_reference.npz"))
    sites = []
    for i in range(int(z["n_sites"])):
        a0, a1, length, tagged = (int(v) for v in z[f"s{i}_span"])
        quals, cigars = z[f"s{i}_quals"], z[f"s{i}_cigars"]
        reads, qp, cp = [], 0, 0
        for bases, n_cig, (rs, mapq, orient, hp) in zip(z[f"s{i}_bases"], z[f"s{i}_n_cigar"], z[f"s{i}_meta"]):
            bases = str(bases)
            reads.append(fo.Read(bases, quals[qp:qp + len(bases)].tolist(),
                                 [(int(o), int(n)) for o, n in cigars[cp:cp + n_cig]], int(rs), int(mapq), int(orient), int(hp)))
            qp, cp = qp + len(bases), cp + int(n_cig)
        sites.append((str(z[f"s{i}_reference"]), a0, a1, length, bool(tagged), reads, z[f"s{i}_expected"]))
    return sites


def test_oracle_equals_the_reference_encoder_on_generated_fixtures():
    sites = reference_generated_sites()
    assert sum(len(s[5]) for s in sites) >= 300
    kinds = set()
    for reference, a0, a1, length, tagged, reads, expected in sites:
        got = fo.features_for_reads(reads, reference, 0, a0, a1, length, include_hp=tagged)
        np.testing.assert_array_equal(got, expected)
        kinds |= {op for r in reads for op, _ in r.cigar}
    assert {fo.BAM_CMATCH, fo.BAM_CINS, fo.BAM_CDEL, fo.BAM_CREF_SKIP, fo.BAM_CEQUAL, fo.BAM_CDIFF} <= kinds
    # the hand-worked answers of this file ARE the reference encoder's answers on its own two cases
    np.testing.assert_array_equal(sites[0][6], EXPECTED)
    np.testing.assert_array_equal(sites[1][6][:, :, :6], EXPECTED)
