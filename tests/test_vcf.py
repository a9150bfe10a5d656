"""Posterior -> genotype -> VCF record: hand-worked cases of every normalisation rule of
the reference (vcfFromContigs.py:139-227, prepareVcf.py:36-105), and the product (hello_amd/vcf.py) against
the literal oracle restatement (oracle/vcf_oracle.py) on random sites."""
import math
import random

import numpy as np
import pytest

from hello_amd import vcf
from oracle import vcf_oracle as vo
from tests.util import canonical_vcf_line, caller_case_sites, load_vcf_reference, reference_segment_onehot

GENOME = "ACGTACGTTTGACCATGCA"


def fields(line):
    chrom, pos, _, ref, alt, qual, flt, info, fmt, gt = line.split("\t")
    return chrom, int(pos), ref, alt, float(qual), flt, info, fmt, gt


def test_snv_het():
    post = {("G", "G"): 0.1, ("G", "T"): 0.8, ("T", "T"): 0.1}
    line = vo.call_alleles(post, "chr1", 2, 1, GENOME)
    assert fields(line) == ("chr1", 3, "G", "T", pytest.approx(-10 * math.log10(0.2), abs=1e-6), "PASS", "HELLO", "GT", "0/1")
    call = vcf.call_site(post, "chr1", 2, 1, GENOME)
    assert call.line() == line


def test_quality_is_capped_at_80():
    post = {("G", "G"): 0.0, ("G", "T"): 1.0, ("T", "T"): 0.0}
    assert fields(vo.call_alleles(post, "c", 2, 1, GENOME))[4] == pytest.approx(80.0, abs=1e-4)
    assert vcf.call_site(post, "c", 2, 1, GENOME).qual == pytest.approx(80.0, abs=1e-4)


@pytest.mark.parametrize("start,ref,alts,want", [
    (3, "T", ["-"], (2, "GT", ["G"])),                 # empty allele: anchor on the previous base
    (3, "TAC", ["TC"], (3, "TA", ["T"])),              # shared suffix trimmed, stops when an allele has 1 base
    (12, "CAT", ["CAG"], (14, "T", ["G"])),            # shared prefix trimmed
    (2, "G", ["GA"], (2, "G", ["GA"])),                # insertion outside a repeat: nothing to trim
    (8, "T", ["TTT"], (6, "G", ["GTT"])),              # insertion in a T run: trimmed and re-anchored until left-aligned
    (8, "TT", ["T"], (6, "GT", ["G"])),                # deletion in a T run: likewise
])
def test_normalisation_rules(start, ref, alts, want):
    got = vcf.normalise(start, ref, alts, GENOME)
    assert (got[0], got[1], list(got[2])) == want
    line = vo.create_vcf_record("c", start, GENOME, ref, alts, [0, 1])
    f = fields(line)
    assert (f[1] - 1, f[2], f[3].split(",")) == want


def test_hom_ref_lists_every_site_allele_and_gt_00():
    post = {("A", "A"): 0.9, ("A", "AT"): 0.05, ("A", "ATT"): 0.01, ("AT", "AT"): 0.02, ("AT", "ATT"): 0.01, ("ATT", "ATT"): 0.01}
    genome = "CCAGG"
    call = vcf.call_site(post, "c", 2, 1, genome)
    assert call.genotype == (0, 0) and call.alts == ("AT", "ATT") and call.ref == "A"
    assert call.line() == vo.call_alleles(post, "c", 2, 1, genome)


def test_site_without_alternative_allele_gives_no_record():
    assert vcf.call_site({("A", "A"): 1.0}, "c", 0, 1, GENOME) is None
    assert vo.call_alleles({("A", "A"): 1.0}, "c", 0, 1, GENOME) is None


def test_two_alt_genotype_indices_follow_sorted_alts():
    post = {("A", "A"): 0.0, ("A", "AT"): 0.1, ("A", "ATT"): 0.1, ("AT", "AT"): 0.1, ("ATT", "AT"): 0.6, ("ATT", "ATT"): 0.1}
    call = vcf.call_site(post, "c", 2, 1, "CCAGG")
    assert call.alts == ("AT", "ATT") and call.genotype == (2, 1)
    assert call.line() == vo.call_alleles(post, "c", 2, 1, "CCAGG")


def test_product_matches_oracle_on_random_sites():
    rng = random.Random(7)
    genome = "".join(rng.choice("ACGT") for _ in range(400))
    n_records = 0
    for _ in range(300):
        start = rng.randrange(5, 380)
        length = rng.choice([1, 1, 2, 3])
        ref = genome[start:start + length]
        alleles = [ref]
        while len(alleles) < rng.choice([2, 2, 3, 4]):
            kind = rng.random()
            if kind < 0.4:
                cand = "".join(rng.choice("ACGT") for _ in range(length))
            elif kind < 0.7:
                cand = ref + "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 2])))
            else:
                cand = ref[:max(0, length - rng.choice([1, 2]))] or "-"
            if cand not in alleles:
                alleles.append(cand)
        rng.shuffle(alleles)
        pairs = [(alleles[i], alleles[j]) for i in range(len(alleles)) for j in range(i, len(alleles))]
        p = np.random.default_rng(rng.randrange(1 << 30)).dirichlet(np.ones(len(pairs)) * 0.3)
        post = dict(zip(pairs, p.tolist()))
        want = vo.call_alleles(post, "chr7", start, length, genome)
        got = vcf.call_site(post, "chr7", start, length, genome)
        assert (got is None) == (want is None)
        if got is not None:
            assert got.line() == want
            n_records += 1
    assert n_records > 200


def test_ensemble_mean_rule():
    e = [{("A", "A"): 0.2, ("A", "T"): 0.7, ("T", "T"): 0.1},
         {("A", "A"): 0.6, ("A", "T"): 0.3, ("T", "T"): 0.1},
         {("A", "A"): 0.1, ("A", "T"): 0.1, ("T", "T"): 0.8}]
    meta = [0.5, 0.3, 0.2]
    mean = vcf.mean_posteriors(e, meta)
    assert mean == vo.mean_of_experts(e, meta)
    assert mean[("A", "T")] == pytest.approx(0.46)
    call = vcf.call_from_prediction((None, e[0], e[1], e[2], meta), "c", 0, 1, "ACGT")
    assert call.genotype == (0, 1) and call.line() == vo.call_alleles(mean, "c", 0, 1, "ACGT")


def test_features_records_round_trip_and_shard_calls_match_oracle(tmp_path):
    """The .features schema (caller_calling.py:743-754,895-898) and prepareVcf's use of it (:126-176)."""
    import pickle
    rng = random.Random(11)
    genomes = {"chr1": "".join(rng.choice("ACGT") for _ in range(300)), "chr2": "".join(rng.choice("ACGT") for _ in range(300))}
    records = []
    for i in range(60):
        chrom = rng.choice(sorted(genomes))
        start = rng.randrange(5, 280)
        ref = genomes[chrom][start]
        alleles = [ref] + rng.sample([b for b in "ACGT" if b != ref], rng.choice([0, 1, 2])) + rng.choice([[], [ref + "G"]])
        pairs = [(alleles[a], alleles[b]) for a in range(len(alleles)) for b in range(a, len(alleles))]
        experts = [dict(zip(pairs, np.random.default_rng(100 * i + k).dirichlet(np.ones(len(pairs)) * 0.4).astype(np.float32)))
                   for k in range(3)]
        meta = np.random.default_rng(i).dirichlet(np.ones(3)).astype(np.float32)
        rec = vcf.feature_record((None, experts[0], experts[1], experts[2], meta), chrom, start, 1)
        assert set(rec) == {"chromosome", "position", "length", "meta", "expertPredictions"}
        assert rec["meta"].dtype == np.float32 and all(isinstance(v, float) for v in rec["expertPredictions"][2].values())
        records.append(rec)
    path = vcf.write_features(str(tmp_path / "shard0.features"), records)
    loaded = pickle.load(open(path, "rb"))                           # what prepareVcf.py:125 does
    got = vcf.calls_from_features(loaded, genomes)
    want = vo.prepare_shard(loaded, genomes)
    line = lambda c: None if c is None else c.line()                 # noqa: E731
    for g, w in zip(list(got.expert) + [got.best, got.mean], want[:5]):
        assert [line(c) for c in g] == w
    assert ["\t".join(map(str, row)) for row in got.choices] == want[5]
    assert sum(c is not None for c in got.mean) > 30


# ---- pins generated by the REFERENCE's own functions (tests/golden/vcf_reference.json, made by
# ---- tests/golden/make_fixtures.py from source ranges of vcfFromContigs.py, prepareVcf.py, caller_calling.py)
def test_record_normalisation_equals_reference_createVcfRecord():
    z = load_vcf_reference()
    n = 0
    for r in z["records"]:
        genome = z["genomes"][r["chromosome"]]
        got = vo.create_vcf_record(r["chromosome"], r["start"], genome, r["ref"], list(r["alts"]), r["gt"], qual=r["qual"])
        assert got == r["line"], r
        norm = vcf.normalise(r["start"], r["ref"], r["alts"], genome)
        if r["line"] is None:
            assert norm is None
        else:
            f = r["line"].split("\t")
            assert (norm[0] + 1, norm[1], ",".join(norm[2])) == (int(f[1]), f[3], f[4])
            n += 1
    assert n > 120


def test_calls_equal_reference_callAlleles():
    z = load_vcf_reference()
    n = 0
    for c in z["calls"]:
        genome = z["genomes"][c["chromosome"]]
        want = canonical_vcf_line(c["line"])
        assert vo.call_alleles(c["likelihoods"], c["chromosome"], c["start"], c["length"], genome) == want
        got = vcf.call_site(c["likelihoods"], c["chromosome"], c["start"], c["length"], genome)
        assert (None if got is None else got.line()) == want
        n += want is not None
    assert n > 180


def test_shard_calls_equal_reference_prepareVcf():
    z = load_vcf_reference()
    sh = z["shard"]
    items = [dict(i, meta=np.asarray(i["meta"], np.float32)) for i in sh["items"]]
    want = [[canonical_vcf_line(x) for x in sh[k]] for k in ("expert0", "expert1", "expert2", "best", "mean")]
    e0, e1, e2, best, mean, choices = vo.prepare_shard(items, z["genomes"])
    assert [e0, e1, e2, best, mean] == want and choices == sh["choices"]
    got = vcf.calls_from_features(items, z["genomes"])
    for g, w in zip(list(got.expert) + [got.best, got.mean], want):
        assert [c.line() for c in g] == w
    assert ["\t".join(map(str, row)) for row in got.choices] == sh["choices"]
    assert len(items) >= 30


@pytest.mark.parametrize("case", [0, 1, 2])
def test_oracle_chain_equals_reference_caller(case):
    """caller_calling.py:612-754 run by the reference on its own network: the oracle chain (per-site wrapper oracle
    -> vcf oracle) must give the same record and the same .features entry for every site."""
    from oracle import moe_oracle as mo
    z = load_vcf_reference()
    spec, state, sites = caller_case_sites(z["caller"][case])
    wrapper = mo.WrapperOracle(spec, state, provide_predictions=True)
    n = 0
    for fd, site in sites:
        genome = z["genomes"][site["chromosome"]]
        seg = reference_segment_onehot(genome, site["start"], site["stop"])
        fd32 = {k: (v[0].astype(np.float32), None if v[1] is None else v[1].astype(np.float32)) for k, v in fd.items()}
        mix, e0, e1, e2, meta = wrapper(fd32, seg.astype(np.float32))
        line = vo.call_alleles({k: float(v) for k, v in mix.items()}, site["chromosome"], site["start"],
                               site["stop"] - site["start"], genome, string="MixtureOfExpertPrediction")
        want = canonical_vcf_line(site["record"])
        assert (line is None) == (want is None)
        if want is None:
            continue
        n += 1
        gf, wf = line.split("\t"), want.split("\t")
        assert gf[:5] + gf[6:] == wf[:5] + wf[6:]                       # everything but QUAL, exactly
        assert abs(float(gf[5]) - float(wf[5])) < 0.05 or float(wf[5]) > 40
        feats = site["features"]
        np.testing.assert_allclose(meta, feats["meta"], rtol=1e-5, atol=1e-6)
        for got_e, want_e in zip((e0, e1, e2), feats["expertPredictions"]):
            assert list(got_e) == list(want_e)                         # same pairs, same order
            np.testing.assert_allclose([float(v) for v in got_e.values()], list(want_e.values()), rtol=1e-4, atol=1e-6)
    assert n >= 11
