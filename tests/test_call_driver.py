"""The call.py-compatible throughput driver (hello_amd/call.py): host side on CPU -- shard files,
the reference's flag surface, FASTA reading, the final-VCF stage -- and, on the GPU, one end-to-end run over
pre-extracted shards against the oracle chain featurizer_oracle -> moe_oracle -> vcf_oracle."""
import os
import pickle

import numpy as np
import pytest

from hello_amd import call, shards, vcf
from hello_amd.featurizer import AlignedRead
from oracle import featurizer_oracle as fo
from oracle import vcf_oracle as vo

REFERENCE_FLAGS = ["--ibam", "--pbam", "--ref", "--workdir", "--chromosomes", "--network", "--hybrid_hotspot",
                   "--q_threshold", "--mapq_threshold", "--num_threads", "--reconcilement_size", "--include_hp"]   # call.py:245-323


def random_read(rng, window_start, ref_len, span, tagged=False):
    """A read with a random CIGAR around the allele span (matches, insertions, deletions, skips, clips)."""
    ops, n_bases, used = [], 0, 0
    if rng.random() < 0.2:
        k = int(rng.integers(1, 6)); ops.append((fo.BAM_CSOFT_CLIP, k)); n_bases += k
    for _ in range(int(rng.integers(1, 5))):
        k = int(rng.integers(5, 70)); ops.append((fo.BAM_CMATCH, k)); n_bases += k; used += k
        u = rng.random()
        if u < 0.2:
            k = int(rng.integers(1, 8)); ops.append((fo.BAM_CINS, k)); n_bases += k
        elif u < 0.4:
            k = int(rng.integers(1, 12)); ops.append((fo.BAM_CDEL, k)); used += k
    k = int(rng.integers(3, 20)); ops.append((fo.BAM_CMATCH, k)); n_bases += k; used += k
    lo = max(window_start + 1, span[0] - used)
    start = min(int(rng.integers(lo, max(lo + 1, span[1]))), window_start + ref_len - used - 1)
    return AlignedRead("".join(rng.choice(list("ACGT"), size=n_bases)), rng.integers(2, 60, size=n_bases).tolist(), ops, start,
                       mapq=int(rng.integers(0, 80)), orientation=int(rng.choice([-1, 1])),
                       hp=int(rng.integers(0, 3)) if tagged else 0)


def random_sites(rng, n, hybrid=False, tagged=False, chromosomes=("chr1", "chr2")):
    sites = []
    for s in range(n):
        window_start = 1000 + 700 * s
        ref_len = 520
        reference = "".join(rng.choice(list("ACGT"), size=ref_len))
        start = window_start + 240 + int(rng.integers(0, 20))
        length = int(rng.choice([1, 1, 2, 3]))
        ref_allele = reference[start - window_start:start - window_start + length]
        names = [ref_allele]
        while len(names) < int(rng.choice([1, 2, 2, 3, 4])):
            u = rng.random()
            cand = ("".join(rng.choice(list("ACGT"), size=length)) if u < 0.4 else
                    ref_allele + "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 3)))) if u < 0.7 else
                    ref_allele[:length - 1])
            if cand not in names:
                names.append(cand)
        names = [names[i] for i in rng.permutation(len(names))]
        alleles = []
        for a in names:
            r0 = [random_read(rng, window_start, ref_len, (start, start + length), tagged) for _ in range(int(rng.choice([0, 1, 3, 6, 11])))]
            r1 = [random_read(rng, window_start, ref_len, (start, start + length)) for _ in range(int(rng.choice([0, 2, 5])))] if hybrid else None
            alleles.append((a, r0, r1))
        sites.append(shards.CandidateSite(chromosomes[s % len(chromosomes)], start, start + length, reference, window_start, alleles))
    return sites


def test_flag_surface_is_the_reference_callers():
    ap = call.parser()
    known = {o for a in ap._actions for o in a.option_strings}
    assert set(REFERENCE_FLAGS) <= known
    args = ap.parse_args(["--network", "m.dnn", "--workdir", "w", "--ibam", "a.bam", "--pbam", "b.bam", "--ref", "r.fa",
                          "--num_threads", "4", "--include_hp", "--mapq_threshold", "5", "--reconcilement_size", "0",
                          "--hybrid_hotspot", "--q_threshold", "7", "--chromosomes", "chr1,chr2"])
    assert args.num_threads == 4 and args.include_hp and args.hybrid_hotspot
    with pytest.raises(SystemExit, match="--shards is required"):
        call.main(ap.parse_args(["--network", "m.dnn", "--workdir", "w", "--ibam", "a.bam"]))
    # features directory named as call.py:33-48 names it
    assert call.features_dir_name("/data/run1/sample.sorted.bam", None) == "features_run1___sample__sorted__bam"
    assert call.features_dir_name(None, None) == "features"


def test_shard_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    for hybrid in (False, True):
        sites = random_sites(rng, 7, hybrid=hybrid, tagged=True)
        back = shards.read_shard(shards.write_shard(str(tmp_path / (f"s{int(hybrid)}.npz" if hybrid else "s.hshard")), sites))
        assert len(back) == len(sites)
        for a, b in zip(sites, back):
            assert (a.chromosome, a.start, a.stop, a.reference, a.window_start) == (b.chromosome, b.start, b.stop, b.reference, b.window_start)
            assert [n for n, _, _ in a.alleles] == [n for n, _, _ in b.alleles]
            for (_, r0, r1), (_, q0, q1) in zip(a.alleles, b.alleles):
                assert [(r.bases, list(r.quals), [tuple(c) for c in r.cigar], r.ref_start, r.mapq, r.orientation, r.hp) for r in r0] == \
                       [(r.bases, list(r.quals), [tuple(c) for c in r.cigar], r.ref_start, min(r.mapq, 255), r.orientation, r.hp) for r in q0]
                assert (r1 is None) == (q1 is None) and (r1 is None or len(r1) == len(q1))


@pytest.mark.parametrize("hybrid,tagged", [(False, False), (True, True)])
def test_packed_shard_arrays_equal_the_unpacked_sites(tmp_path, hybrid, tagged):
    """The driver keeps a shard as the flat arrays of its file (shards.PackedShard): the featurizer arrays it derives by
    index arithmetic -- dummy reads for unsupported alleles included -- are element for element what pack_sites builds
    from the unpacked AlignedRead objects, and the per-site metadata is the sites'."""
    from hello_amd import featurizer
    rng = np.random.default_rng(31)
    sites = random_sites(rng, 60, hybrid=hybrid, tagged=tagged)
    assert any(len(r0) == 0 for s in sites for _, r0, _ in s.alleles)          # alleles without supporting reads
    path = shards.write_shard(str(tmp_path / "s.hshard"), sites)
    for packed in (shards.PackedShard.from_file(path), shards.PackedShard.from_sites(sites)):
        assert len(packed) == len(sites) and packed.hybrid == hybrid and packed.has_reads(1) == hybrid
        for tech in ((0, 1) if hybrid else (0,)):
            want = featurizer.pack_sites([s.site_reads(tech) for s in sites])
            got = packed.featurizer_arrays(tech)
            assert set(got) == set(want)
            for k in want:
                assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
        for i, s in enumerate(sites):
            assert (packed.chromosomes[i], int(packed.start[i]), int(packed.stop[i]), int(packed.window_start[i])) == \
                   (s.chromosome, s.start, s.stop, s.window_start)
            assert packed.names(i) == [a for a, _, _ in s.alleles] and packed.reference(i) == s.reference


def test_fasta_and_window_reference(tmp_path):
    path = tmp_path / "g.fa"
    path.write_text(">chr1 first\nACGTAC\nGTTT\n>chr2\nGGGG\nCC\n>chrUn\nNNNN\n")
    g = call.read_fasta(str(path), ["chr1", "chr2"])
    assert g == {"chr1": "ACGTACGTTT", "chr2": "GGGGCC"}
    w = call.WindowReference("ACGTACGTTT", 100)
    assert w[102:105] == "GTA" and w[109] == "T"
    with pytest.raises(IndexError):
        w[99]
    seg = call.reference_segment("ACGTN" * 40, 100, 101, span=150)
    assert seg.shape == (150, 5) and seg.sum() == 150
    assert seg[0].tolist() == [1, 0, 0, 0, 0] and seg[4].tolist() == [0, 0, 0, 0, 1]      # window [25, 175): A ... N


def test_final_vcf_stage_equals_the_oracle_of_prepareVcf(tmp_path):
    """prepare_vcf on .features files: the meta-weighted-mean calls of oracle.vcf_oracle.prepare_shard (itself
    pinned by the reference's prepareVcf.vcfRecords output), sorted, under the reference's header."""
    from tests.util import load_vcf_reference
    z = load_vcf_reference()
    items = [dict(i, meta=np.asarray(i["meta"], np.float32)) for i in z["shard"]["items"]]
    half = len(items) // 2
    files = [vcf.write_features(str(tmp_path / f"features{k}.features"), part) for k, part in enumerate((items[half:], items[:half]))]
    lengths = {c: len(g) for c, g in z["genomes"].items()}
    out = call.prepare_vcf(files, str(tmp_path / "results.output.vcf"), lambda rec: z["genomes"][rec["chromosome"]], lengths)
    lines = open(out).read().split("\n")[:-1]
    head = [ln for ln in lines if ln.startswith("#")]
    body = [ln for ln in lines if not ln.startswith("#")]
    assert head[0] == "##fileformat=VCFv4.1" and head[-1].startswith("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT")
    assert "##contig=<ID=chrA,length=%d>" % lengths["chrA"] in head
    want = vo.prepare_shard(items, z["genomes"])[4]
    assert sorted(body) == sorted(want)
    keys = [(ln.split("\t")[0], int(ln.split("\t")[1])) for ln in body]
    assert keys == sorted(keys)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,hybrid,tagged", [("single_tech", False, False), ("single_tech_hp", False, True),
                                               ("hybrid_ensemble2", True, False)])
def test_driver_end_to_end_matches_the_oracle_chain(tmp_path, cfg, hybrid, tagged):
    """python -m hello_amd.call over two pre-extracted shards: per-shard .vcf + .features + sentinel logs + the final
    VCF, against featurizer_oracle -> moe_oracle (per-site wrapper) -> vcf_oracle on the same sites."""
    from hello_amd import loader, netspec as ns, weights
    from oracle import moe_oracle as mo
    spec = ns.build(cfg)
    state = weights.synth_state(spec, seed=17)
    model = str(tmp_path / "model.hello.npz")
    loader.save_native(model, cfg, state)
    rng = np.random.default_rng(23)
    shard_sites = [random_sites(rng, 30, hybrid, tagged), random_sites(rng, 17, hybrid, tagged)]
    os.makedirs(tmp_path / "shards")
    for k, sites in enumerate(shard_sites):
        shards.write_shard(str(tmp_path / "shards" / (f"shard{k}.npz" if k else f"shard{k}.hshard")), sites)
    argv = ["--network", model, "--workdir", str(tmp_path / "work"), "--shards", str(tmp_path / "shards"), "--num_threads", "2"]
    if tagged:
        argv.append("--include_hp")
    result = call.main(call.parser().parse_args(argv))
    fdir = tmp_path / "work" / "features"
    wrapper = mo.WrapperOracle(spec, state, provide_predictions=True)
    n_records, mean_lines = 0, []
    for k, sites in enumerate(shard_sites):
        assert call.SENTINEL in open(fdir / f"features{k}.log").read()
        got_lines = open(fdir / f"features{k}.vcf").read().split("\n")[:-1]
        got_feats = pickle.load(open(fdir / f"features{k}.features", "rb"))
        assert len(got_lines) == len(got_feats)
        at = 0
        for site in sites:
            genome = call.WindowReference(site.reference, site.window_start)
            fd = {}
            for name, r0, r1 in site.alleles:
                def enc(reads, hp):
                    return fo.features_for_reads([fo.Read(r.bases, r.quals, r.cigar, r.ref_start, r.mapq, r.orientation, r.hp) for r in reads],
                                                 site.reference, site.window_start, site.start, site.stop, 150, hp).astype(np.float32)
                fd[name] = (enc(r0, tagged), enc(r1, False) if hybrid else None)
            seg = call.reference_segment(genome, site.start, site.stop)[None].astype(np.float32)
            mix, e0, e1, e2, meta = wrapper(fd, seg)
            want = vo.call_alleles({k2: float(v) for k2, v in mix.items()}, site.chromosome, site.start, site.stop - site.start,
                                   genome, string="MixtureOfExpertPrediction")
            if want is None:
                continue
            line, feats = got_lines[at], got_feats[at]
            at += 1
            n_records += 1
            assert (feats["chromosome"], feats["position"], feats["length"]) == (site.chromosome, site.start, site.stop - site.start)
            np.testing.assert_allclose(feats["meta"], meta, atol=1e-4)
            for ge, we in zip(feats["expertPredictions"], (e0, e1, e2)):
                assert list(ge) == list(we)
                assert np.abs(np.array(list(ge.values())) - np.array([float(v) for v in we.values()])).max() < 1e-4
            ps = sorted((float(v) for v in mix.values()), reverse=True)
            if len(ps) > 1 and ps[0] - ps[1] < 2e-4:
                continue                                    # a genuinely ambiguous site may flip
            gf, wf = line.split("\t"), want.split("\t")
            assert gf[:5] + gf[6:] == wf[:5] + wf[6:], (line, want)
            if 1.0 - ps[0] > 1e-2:
                assert abs(float(gf[5]) - float(wf[5])) < 0.05
        assert at == len(got_lines)
    assert n_records >= 30
    body = [ln for ln in open(result).read().split("\n")[:-1] if not ln.startswith("#")]
    assert len(body) >= 30 and all(ln.split("\t")[7] == "HELLO" for ln in body)


def _write_model_and_shards(tmp_path, cfg, n_shards, sites_per_shard, hybrid=False, tagged=False, seed=5):
    from hello_amd import loader, netspec as ns, weights
    spec = ns.build(cfg)
    model = str(tmp_path / "model.hello.npz")
    loader.save_native(model, cfg, weights.synth_state(spec, seed=17))
    rng = np.random.default_rng(seed)
    os.makedirs(tmp_path / "shards", exist_ok=True)
    for k in range(n_shards):
        n = sites_per_shard if isinstance(sites_per_shard, int) else sites_per_shard[k]
        sites = random_sites(rng, n, hybrid, tagged, chromosomes=("chr%d" % (1 + k % 3), "chr2"))
        for s in sites:                                     # spread the shards' positions so the final sort interleaves them
            s.start += 50_000 * (k % 4); s.stop += 50_000 * (k % 4); s.window_start += 50_000 * (k % 4)       # noqa: E702
            for _, r0, r1 in s.alleles:
                for rd in r0 + (r1 or []):
                    rd.ref_start += 50_000 * (k % 4)
        shards.write_shard(str(tmp_path / "shards" / (f"shard{k}.npz" if k == 2 else f"shard{k}.hshard")), sites)   # one NumPy archive among them
    return model


def _files(directory):
    return {name: open(os.path.join(directory, name), "rb").read() for name in sorted(os.listdir(directory))
            if not name.endswith(".log") and not name.startswith("mean_index")}


def _same_calls(a, b, p_tol=2e-6):
    """Two runs' files hold the same records: strings equal, probabilities within ``p_tol`` (a site's posteriors depend at
    rounding level on the launch it was part of -- the per-allele read sums are cut where workgroups end), QUAL
    accordingly."""
    assert sorted(a) == sorted(b)
    for name in a:
        if name.endswith(".features"):
            fa, fb = pickle.loads(a[name]), pickle.loads(b[name])
            assert len(fa) == len(fb)
            for x, y in zip(fa, fb):
                assert (x["chromosome"], x["position"], x["length"]) == (y["chromosome"], y["position"], y["length"])
                assert np.abs(x["meta"] - y["meta"]).max() <= p_tol
                for dx, dy in zip(x["expertPredictions"], y["expertPredictions"]):
                    assert list(dx) == list(dy)
                    assert np.abs(np.array(list(dx.values())) - np.array(list(dy.values()))).max() <= p_tol
        else:
            la, lb = a[name].decode().split("\n"), b[name].decode().split("\n")
            assert len(la) == len(lb), name
            for x, y in zip(la, lb):
                fx, fy = x.split("\t"), y.split("\t")
                if len(fx) > 5 and fx != fy:
                    p = 1.0 - 10 ** (-float(fx[5]) / 10)             # d QUAL = 4.34 dp / (1 - p)
                    assert fx[:5] + fx[6:] == fy[:5] + fy[6:] and abs(float(fx[5]) - float(fy[5])) <= 4.4 * p_tol / max(1 - p, 1e-8) + 1e-5, (x, y)


@pytest.mark.gpu
def test_launch_coalescing_and_pipeline_depth_are_invisible(tmp_path):
    """The same shards as one launch each, coalesced three at a time and all in one launch: the same records in every
    per-shard file and in the final VCF (the launch is not the file: offsets, site indices and CSR counts are renumbered
    per launch); an empty shard still completes with its sentinel; the same launches twice are the same bytes."""
    model = _write_model_and_shards(tmp_path, "hybrid_ensemble2", 7, [9, 0, 14, 5, 11, 1, 8], hybrid=True)
    outs = []
    for label, per_launch in (("one", 1), ("three", 30), ("all", 100000), ("again", 30)):
        work = tmp_path / label
        result = call.main(call.parser().parse_args(["--network", model, "--workdir", str(work), "--shards", str(tmp_path / "shards"),
                                                     "--num_threads", "4", "--sites_per_launch", str(per_launch)]))
        files = _files(work / "features")
        files["results.output.vcf"] = open(result, "rb").read()
        outs.append(files)
        assert call.SENTINEL in open(work / "features" / "features1.log").read()
        assert pickle.load(open(work / "features" / "features1.features", "rb")) == []
    _same_calls(outs[0], outs[1])
    _same_calls(outs[0], outs[2])
    assert outs[1] == outs[3]
    assert len(outs[0]) == 7 * 3 + 1 and outs[0]["results.output.vcf"].count(b"\n") > 20


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_write_the_single_process_result(tmp_path):
    """``--gpus 2`` re-launches the command under torch.distributed.run: two ranks (sharing this box's one GPU) are dealt
    the shards by read count, write their shards' files, meet at one barrier, and rank 0 merges the final VCF.  With one
    shard per launch both runs issue the same launches: every file and the final VCF are byte for byte the single
    process's; with the default coalescing the launches differ between the runs and the records agree (``_same_calls``)."""
    import subprocess
    import sys
    model = _write_model_and_shards(tmp_path, "single_tech", 9, [12, 30, 7, 22, 3, 16, 25, 9, 14])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for label, extra in (("exact", ["--sites_per_launch", "1"]), ("coalesced", [])):
        single = call.main(call.parser().parse_args(["--network", model, "--workdir", str(tmp_path / f"single_{label}"),
                                                     "--shards", str(tmp_path / "shards"), "--num_threads", "4"] + extra))
        run = subprocess.run([sys.executable, "-m", "hello_amd.call", "--network", model, "--workdir", str(tmp_path / f"two_{label}"),
                              "--shards", str(tmp_path / "shards"), "--num_threads", "4", "--gpus", "2"] + extra,
                             env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-3000:]
        assert "rank 1 of 2" in run.stderr and "rank 0 of 2" in run.stderr
        one = dict(_files(tmp_path / f"single_{label}" / "features"), final=open(single, "rb").read())
        two = dict(_files(tmp_path / f"two_{label}" / "features"), final=open(tmp_path / f"two_{label}" / "results.output.vcf", "rb").read())
        if label == "exact":
            assert one == two
        else:
            _same_calls(one, two)
        assert one["final"].count(b"\n") > 40
