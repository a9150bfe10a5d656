"""The host-side record stage of the library (hello_site_records, include/hello_mi355x.h; hello_amd/records.py) on CPU:
against the readable rules of hello_amd/vcf.py on random sites (ties, empty / '-' alleles, sites without the reference
allele, several shards per call), against the REFERENCE's own outputs (tests/golden/vcf_reference.json: prepareVcf.callAlleles
cases and one .features shard through prepareVcf.vcfRecords), its pickle streams, its errors; the final-VCF merge; the
validation of shard files."""
import pickle
import pickletools

import numpy as np
import pytest

from hello_amd import records as R, shard_pipeline as sp, shards, vcf
from hello_amd.wrapper import pair_keys
from oracle import vcf_oracle as vo
from tests.test_call_driver import random_sites
from tests.util import canonical_vcf_line, load_vcf_reference


def random_table(rng, S, genome, names=("chr1", "chrX")):
    aps, alleles, starts, stops = [], [], [], []
    for s in range(S):
        start = 500 + 50 * s + int(rng.integers(0, 5))
        length = int(rng.choice([0, 1, 1, 2, 3]))
        ref = genome[start:start + length]
        n = int(rng.choice([1, 2, 2, 3, 4]))
        al = [ref]
        while len(al) < n:
            u = rng.random()
            c = ("".join(rng.choice(list("ACGT"), size=max(length, 1))) if u < 0.4 else
                 ref + "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 3)))) if u < 0.7 else
                 ref[:max(length - 1, 0)] if u < 0.9 else ref + "-")
            if c not in al:
                al.append(c)
        if rng.random() < 0.1 and n > 1:
            al = al[1:]                                   # a site that does not list the reference allele
        al = [al[i] for i in rng.permutation(len(al))]
        aps.append(len(al)); alleles += al; starts.append(start); stops.append(start + length)       # noqa: E702
    aps = np.array(aps, np.int32)
    P = int((aps * (aps + 1) // 2).sum())
    post = rng.random((4, P)).astype(np.float32)
    post[:, ::7] = 0.5                                    # equal probabilities: broken by the pair's strings
    post[0, ::11] = 1.0                                   # QUAL cap
    meta = rng.dirichlet([1, 1, 1], size=S).astype(np.float32)
    chrom = rng.integers(0, len(names), size=S)
    return aps, alleles, np.array(starts), np.array(stops), post, meta, chrom


@pytest.mark.parametrize("threads,with_meta", [(1, True), (4, False)])
def test_record_stage_equals_the_python_rules(threads, with_meta):
    rng = np.random.default_rng(11 + threads)
    S = 1500
    genome = "".join(rng.choice(list("ACGT"), size=S * 50 + 1000))
    names = ["chr1", "chrX"]
    aps, alleles, starts, stops, post, meta, chrom = random_table(rng, S, genome)
    if not with_meta:
        meta = None
    keep = (rng.random(S) < 0.9).astype(np.uint8)
    text, off = R.text_table(np.array(alleles))
    table = R.SiteTable(aps, text, off, names, chrom, starts, stops, genomes={n: genome for n in names}, keep=keep)
    cuts = [0, 400, 400, 1100, S]
    with R.site_records(table, post, meta, shard_site_off=cuts, threads=threads) as rec:
        col = a0 = 0
        want_features = []
        for s in range(S):
            al = alleles[a0:a0 + aps[s]]; a0 += aps[s]                                              # noqa: E702
            keys = pair_keys(al)
            rows = [dict(zip(keys, post[r, col:col + len(keys)].astype(np.float64).tolist())) for r in range(4)]
            col += len(keys)
            m = meta[s] if meta is not None else np.array([1, 0, 0], np.float32)
            call = vcf.call_site(rows[0], names[chrom[s]], int(starts[s]), int(stops[s] - starts[s]), genome, info="MixtureOfExpertPrediction")
            if not keep[s]:
                call = None
            got = bytes(rec.shard_vcf[rec.shard_vcf_off[s]:rec.shard_vcf_off[s + 1]]).decode()
            assert got == (call.line() + "\n" if call else ""), s
            best_p, best_pair = max((p, pair) for pair, p in rows[0].items())
            assert keys[rec.best_pair[0, s]] == best_pair and rec.best_p[0, s] == best_p
            got_mean = bytes(rec.mean_vcf[rec.mean_vcf_off[s]:rec.mean_vcf_off[s + 1]]).decode()
            if call is None:
                assert got_mean == "" and rec.mean_position[s] == -1
                continue
            entry = vcf.feature_record((rows[0], rows[1], rows[2], rows[3], m), names[chrom[s]], int(starts[s]), int(stops[s] - starts[s]))
            want_features.append(entry)
            mean = vcf.call_site(vcf.mean_posteriors(entry["expertPredictions"], entry["meta"]), names[chrom[s]], int(starts[s]),
                                 int(stops[s] - starts[s]), genome)
            assert got_mean == (mean.line() + "\n" if mean else ""), s        # ("X-" normalises to X: possibly no record)
            if mean:
                assert rec.mean_position[s] == mean.position and rec.qual[4, s] == mean.qual
        got_features = []
        for k in range(len(cuts) - 1):
            stream = bytes(rec.features[rec.features_off[k]:rec.features_off[k + 1]])
            assert sum(1 for _ in pickletools.genops(stream)) > 0            # a well-formed opcode stream
            part = pickle.loads(stream)
            assert len(part) == rec.n_records[k]
            got_features += part
        assert pickle.loads(bytes(rec.features[rec.features_off[1]:rec.features_off[2]])) == []     # the empty shard
        assert len(got_features) == len(want_features) > S // 2
        for g, w in zip(got_features, want_features):
            assert (g["chromosome"], g["position"], g["length"]) == (w["chromosome"], w["position"], w["length"])
            assert isinstance(g["meta"], np.ndarray) and g["meta"].dtype == np.float32 and np.array_equal(g["meta"], w["meta"])
            assert g["expertPredictions"] == w["expertPredictions"]
            assert [list(d) for d in g["expertPredictions"]] == [list(d) for d in w["expertPredictions"]]       # pair order


def test_features_stream_bytes_do_not_depend_on_the_worker_count():
    """a shard's ``.features`` stream is ONE MARK ... APPENDS group whatever worker chunks its records came from, so
    the file is byte-reproducible across --num_threads and hosts (the VCF texts always were)."""
    rng = np.random.default_rng(5)
    S = 1300
    genome = "".join(rng.choice(list("ACGT"), size=S * 50 + 1000))
    names = ["chr1", "chrX"]
    aps, alleles, starts, stops, post, meta, chrom = random_table(rng, S, genome)
    text, off = R.text_table(np.array(alleles))
    table = R.SiteTable(aps, text, off, names, chrom, starts, stops, genomes={n: genome for n in names})
    cuts = [0, 3, 3, 700, 1290, S]
    streams = {}
    for threads in (1, 2, 4, 8):
        with R.site_records(table, post, meta, shard_site_off=cuts, threads=threads) as rec:
            streams[threads] = (bytes(rec.features), rec.features_off.copy(), bytes(rec.shard_vcf), bytes(rec.mean_vcf), rec.n_records.copy())
    for threads in (2, 4, 8):
        for a, b in zip(streams[1], streams[threads]):
            assert (a == b) if isinstance(a, bytes) else np.array_equal(a, b), threads
    feats, foff = streams[1][0], streams[1][1]
    for k in range(len(cuts) - 1):
        stream = feats[foff[k]:foff[k + 1]]
        ops = [op.name for op, _, _ in pickletools.genops(stream)]
        assert ops.count("APPENDS") == (1 if cuts[k + 1] > cuts[k] and streams[1][4][k] else 0)
        assert len(pickle.loads(stream)) == streams[1][4][k]


def _single_site_table(chromosome, start, length, alleles, genome):
    text, off = R.text_table(np.array(alleles))
    return R.SiteTable([len(alleles)], text, off, [chromosome], [0], [start], [start + length], genomes={chromosome: genome})


def test_record_stage_reproduces_the_reference_callAlleles_cases():
    """vcf_reference.json 'calls': prepareVcf.callAlleles executed by the reference on seeded inputs.  The library takes
    float32 posteriors, so every case is held to the pinned oracle on the float32-rounded likelihoods, and -- whenever the
    rounding leaves the printed QUAL alone -- to the reference's own line."""
    z = load_vcf_reference()
    direct = 0
    for case in z["calls"]:
        pairs = list(case["likelihoods"])
        alleles = []
        for pair in pairs:
            for a in pair:
                if a not in alleles:
                    alleles.append(a)
        if pairs != pair_keys(alleles):
            continue                                        # not in the wrapper's pair order: the per-site path covers it
        values = np.array([case["likelihoods"][p] for p in pairs], np.float32)
        post = np.stack([values] * 4)
        genome = z["genomes"][case["chromosome"]]
        table = _single_site_table(case["chromosome"], case["start"], case["length"], alleles, genome)
        with R.site_records(table, post, None, features=False) as rec:
            got = bytes(rec.mean_vcf).decode().rstrip("\n") or None        # meta [1, 0, 0]: the mean row is expert 0 in float64
        rounded = {p: float(v) for p, v in zip(pairs, values)}
        assert got == canonical_vcf_line(vo.call_alleles(rounded, case["chromosome"], case["start"], case["length"], genome))
        if got == canonical_vcf_line(case["line"]):
            direct += 1
    assert direct >= 150


def test_record_stage_reproduces_the_reference_final_stage_on_a_features_shard():
    """vcf_reference.json 'shard': a .features list pushed through the reference's prepareVcf.vcfRecords -- its mean.vcf
    lines (the final VCF's) and per-expert decisions against the library's, from the same float32 posteriors."""
    z = load_vcf_reference()
    sh = z["shard"]
    items = sh["items"]
    alleles, aps, starts, stops, chrom, cols = [], [], [], [], [], []
    names = sorted({i["chromosome"] for i in items})
    for it in items:
        pairs = list(it["expertPredictions"][0])
        al = []
        for pair in pairs:
            for a in pair:
                if a not in al:
                    al.append(a)
        assert pairs == pair_keys(al)
        alleles += al; aps.append(len(al)); starts.append(it["position"]); stops.append(it["position"] + it["length"])      # noqa: E702
        chrom.append(names.index(it["chromosome"]))
        cols.append(np.array([[it["expertPredictions"][e][p] for p in pairs] for e in range(3)]))
    experts = np.concatenate(cols, axis=1)
    assert np.array_equal(experts.astype(np.float32).astype(np.float64), experts)       # the fixture holds float32 values
    meta = np.array([it["meta"] for it in items], np.float32)
    mix = (experts * meta.T[:, np.repeat(np.arange(len(items)), [a * (a + 1) // 2 for a in aps])]).sum(0)
    post = np.concatenate([mix[None], experts]).astype(np.float32)
    text, off = R.text_table(np.array(alleles))
    table = R.SiteTable(aps, text, off, names, chrom, starts, stops, genomes=z["genomes"])
    with R.site_records(table, post, meta, features=False) as rec:
        for s, it in enumerate(items):
            got = bytes(rec.mean_vcf[rec.mean_vcf_off[s]:rec.mean_vcf_off[s + 1]]).decode().rstrip("\n") or None
            if bytes(rec.shard_vcf[rec.shard_vcf_off[s]:rec.shard_vcf_off[s + 1]]):
                assert got == canonical_vcf_line(sh["mean"][s]), s
            for e in range(3):                              # the experts' decisions: best pair and QUAL of expert<e>.vcf
                line = sh[f"expert{e}"][s]
                if line is not None:
                    assert abs(float(line.split("\t")[5]) - rec.qual[1 + e, s]) < 1e-6
    assert len(items) >= 20


def test_record_stage_errors_and_strings():
    text, off = R.text_table(np.array(["A", "", "ACGT" * 80, "T-"]))
    assert bytes(text[:-1]).decode() == "A" + "ACGT" * 80 + "T-" and off.tolist() == [0, 1, 1, 321, 323]
    assert R.text_table(np.array([], "U1"))[1].tolist() == [0]
    prefix, suffix = R.meta_pickle_format()
    arr = pickle.loads(b"\x80\x03" + prefix + np.array([0.25, 0.5, 0.25], np.float32).tobytes() + suffix + b".")
    assert arr.dtype == np.float32 and arr.tolist() == [0.25, 0.5, 0.25]
    # a site whose normalisation needs a base left of its reference window: an error naming it, not a wrong record
    aps = [2]
    text, off = R.text_table(np.array(["", "C"]))
    window = np.frombuffer(b"CGGT", np.uint8)
    table = R.SiteTable(aps, text, off, ["chr1"], [0], [100], [101], ref_windows=window, ref_window_off=[0, 4], window_start=[100])
    post = np.array([[0.1, 0.8, 0.1]] * 4, np.float32)
    with pytest.raises(RuntimeError, match="chr1:100.*outside the reference available"):
        R.site_records(table, post, None)
    table = R.SiteTable(aps, text, off, ["chr1"], [0], [100], [101], ref_windows=window, ref_window_off=[0, 4], window_start=[99])
    with R.site_records(table, post, None) as rec:
        assert bytes(rec.shard_vcf).decode() == "chr1\t100\t.\tCG\tC,CC\t6.989700\tPASS\tMixtureOfExpertPrediction\tGT\t1/2\n"
    with pytest.raises(RuntimeError, match="n_pairs_total"):
        R.site_records(table, post[:, :2], None)
    with pytest.raises(ValueError, match="one entry per site"):
        R.SiteTable(aps, text, off, ["chr1"], [0, 0], [100], [101], ref_windows=window, ref_window_off=[0, 4], window_start=[99])


def test_final_vcf_merge_sorts_by_chromosome_name_then_position(tmp_path):
    rng = np.random.default_rng(3)
    outputs, everything = [], []
    for k in range(7):
        names = [["chr2", "chr10", "chr1"], ["chrX"], ["chr1", "chr2"]][k % 3]
        n = int(rng.integers(0, 40)) if k != 3 else 0
        chrom_of = rng.integers(0, len(names), size=n).astype(np.int32)
        pos = rng.integers(0, 500, size=n).astype(np.int64)
        lines = [f"{names[c]}\t{p + 1}\t.\tA\t{'T' * int(rng.integers(1, 9))}\t1.0\tPASS\tHELLO\tGT\t0/1\n".encode() for c, p in zip(chrom_of, pos)]
        prefix = str(tmp_path / f"features{k}")
        open(prefix + ".mean.vcf", "wb").write(b"".join(lines))
        outputs.append(sp.ShardOutput(k, prefix, n, n, names, chrom_of, pos, np.array([len(ln) for ln in lines], np.int64)))
        everything += [(names[c], int(p), k, i, ln) for i, (c, p, ln) in enumerate(zip(chrom_of, pos, lines))]
    out = str(tmp_path / "final.vcf")
    n = sp.merge_final_vcf(outputs, lambda names: "#" + ",".join(names) + "\n", out)
    want = b"".join(ln for *_, ln in sorted(everything, key=lambda t: t[:4]))
    got = open(out, "rb").read()
    assert n == len(everything) and got == b"#chr1,chr10,chr2,chrX\n" + want
    # the same through a rank's saved index
    again = sp.load_index(sp.save_index(str(tmp_path / "index.npz"), outputs))
    assert sp.merge_final_vcf(again, lambda names: "#" + ",".join(names) + "\n", out) == n and open(out, "rb").read() == got
    assert sp.merge_final_vcf([], lambda names: "#\n", out) == 0 and open(out, "rb").read() == b"#\n"


def test_malformed_shards_are_refused(tmp_path):
    """everything the featurizer kernel indexes with is validated when a shard is loaded."""
    rng = np.random.default_rng(9)
    sites = random_sites(rng, 12, hybrid=True, tagged=True)
    good = shards._payload(sites)
    shards.PackedShard(dict(good))                                           # loads
    for name, kw in (("s.hshard", {}), ("s.npz", {}), ("c.npz", dict(compressed=True))):
        back = shards.PackedShard.from_file(shards.write_shard(str(tmp_path / name), sites, **kw))
        assert back.n_sites == 12 and back.allele_names == [a for s in sites for a, _, _ in s.alleles]
        assert back.chromosomes == [s.chromosome for s in sites]
        for k, v in good.items():
            assert np.array_equal(back.z[k], v), (name, k)
    with pytest.raises(ValueError, match="not a shard file"):
        open(tmp_path / "junk.hshard", "wb").write(b"PK\x03\x04" + bytes(40))
        shards.PackedShard.from_file(str(tmp_path / "junk.hshard"))

    def broken(**change):
        z = dict(good)
        z.update(change)
        return z
    cases = {
        "ref_off ends": broken(ref=good["ref"][:-3]),
        "does not cover the feature window": broken(window_start=good["window_start"] + np.where(np.arange(12) == 5, 300, 0)),
        "consumes": broken(cigars0=np.where(np.arange(good["cigars0"].shape[0]) == 2, good["cigars0"] + (1 << 4), good["cigars0"])),
        "read_off0 ends": broken(bases0=good["bases0"][:-1], quals0=good["quals0"][:-1]),
        "cigar_off1 ends": broken(cigars1=good["cigars1"][:-1]),
        "quals0 and bases0": broken(quals0=good["quals0"][:-1]),
        "allele strings": broken(allele_text_off=good["allele_text_off"][:-1], allele_text=good["allele_text"][:int(good["allele_text_off"][-2])]),
        "offsets of the allele strings": broken(allele_text=good["allele_text"][:-1]),
        "outside the chromosome names": broken(chromosome_of_site=good["chromosome_of_site"] + 2),
        "reads_per_allele0 holds": broken(reads_per_allele0=good["reads_per_allele0"][:-1]),
        "mapq0 holds": broken(mapq0=good["mapq0"][:-1]),
        "stop < start": broken(stop=good["stop"] - np.where(np.arange(12) == 7, 50, 0)),
        "a site without alleles": broken(alleles_per_site=np.where(np.arange(12) == 0, 0, good["alleles_per_site"]).astype(np.int32)),
    }
    for message, arrays in cases.items():
        with pytest.raises(ValueError, match=message):
            shards.PackedShard(arrays)
    with pytest.raises(ValueError, match=r"site 5, chr\d:\d+"):             # the site is named
        shards.PackedShard(cases["does not cover the feature window"])
    # a file that says has_second = 0 but carries arrays of a second technology used to pass validation (only the
    # technologies the flag names were checked) while has_reads(1) still said yes to a hybrid model -- unvalidated offsets on the GPU
    bogus = broken(has_second=np.array(0), read_off1=np.array([0, 10 ** 9, 2 * 10 ** 9], np.int64))
    with pytest.raises(ValueError, match="has_second is 0 but the file carries arrays of a second technology"):
        shards.PackedShard(bogus)
    single = shards.PackedShard(shards._payload(random_sites(rng, 5)))
    assert single.has_reads(0) and not single.has_reads(1) and single.n_reads(1) == 0
    with pytest.raises(ValueError, match=r"arrays of technology 1 are missing \(cigars"):
        shards.PackedShard({k: v for k, v in good.items() if k != "cigars1"})

    # the header of a .hshard file is the file's own claim -- negative offsets / dimensions, foreign dtypes and
    # truncated arrays are refused by name, in the whole-file reader and in the count-arrays-only reader alike
    import json as _json
    path = shards.write_shard(str(tmp_path / "h.hshard"), sites)
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[8:16], np.uint64)[0])
    header = _json.loads(raw[16:16 + n])

    def rewritten(name, **patch):
        h = {k: list(v) for k, v in header.items()}
        for k, (field, value) in patch.items():
            h[k][field] = value
        head = _json.dumps(h).encode("ascii")           # array offsets count from the end of the header, which is padded so that
        head += b" " * ((-(16 + len(head))) % 64)       # the arrays keep their absolute alignment (write_flat does the same)
        out = str(tmp_path / name)
        open(out, "wb").write(raw[:8] + np.uint64(len(head)).tobytes() + head + raw[16 + n:])
        return out
    attacks = {
        "negative offset or dimension": rewritten("neg_at.hshard", start=(2, -8)),
        "negative offset or dimension.": rewritten("neg_dim.hshard", reads_per_allele0=(1, [-3])),
        "integer or float arrays only": rewritten("obj.hshard", start=(0, "|O")),
        "runs past the end of the file": rewritten("long.hshard", reads_per_allele0=(1, [10 ** 9])),
    }
    for message, bad_path in attacks.items():
        with pytest.raises(ValueError, match=message.rstrip(".")):
            shards.read_flat(bad_path)
        if "start" not in message and bad_path.endswith(("neg_dim.hshard", "long.hshard")):
            with pytest.raises(ValueError, match=message.rstrip(".")):
                shards.read_flat_arrays(bad_path, ["reads_per_allele0"])
    with pytest.raises(ValueError, match="negative offset"):
        shards.read_flat_arrays(attacks["negative offset or dimension"], ["start"])
    open(tmp_path / "cut.hshard", "wb").write(raw[:16 + n // 2])
    for reader in (shards.read_flat, lambda p: shards.read_flat_arrays(p, ["start"])):
        with pytest.raises(ValueError, match="header is cut short"):
            reader(str(tmp_path / "cut.hshard"))
    assert np.array_equal(shards.read_flat_arrays(path, ["start", "nothing"])["start"], good["start"])
    # JSON numbers that are not integers (int() would truncate 1.9 to a plausible 1, True to 1), offsets that are not
    # aligned to the element (the view goes by pointer to C and the GPU), arrays that share bytes
    more = {
        "non-integer offset or dimension": rewritten("frac_dim.hshard", reads_per_allele0=(1, [1.9])),
        "non-integer offset or dimension.": rewritten("bool_dim.hshard", reads_per_allele0=(1, [True])),
        "non-integer offset or dimension..": rewritten("frac_at.hshard", start=(2, 64.0)),
        "not a multiple of its 8-byte element": rewritten("odd_at.hshard", start=(2, header["start"][2] + 4)),
        "overlap": rewritten("overlap.hshard", stop=(2, header["start"][2])),
    }
    for message, bad_path in more.items():
        with pytest.raises(ValueError, match=message.rstrip(".")):
            shards.read_flat(bad_path)
    for bad in ("frac_dim.hshard", "bool_dim.hshard", "overlap.hshard"):
        with pytest.raises(ValueError, match="non-integer|overlap"):
            shards.read_flat_arrays(str(tmp_path / bad), ["reads_per_allele0"])
    # alignment is a property of the ABSOLUTE position: a foreign writer's header of odd length (write_flat pads to 64)
    # shifts every array off its element's alignment although each offset alone looks aligned
    head = _json.dumps(header).encode("ascii")
    head += b" " * ((-(16 + len(head))) % 64 + 3)
    open(tmp_path / "odd_header.hshard", "wb").write(raw[:8] + np.uint64(len(head)).tobytes() + head + raw[16 + n:])
    for reader in (shards.read_flat, lambda p: shards.read_flat_arrays(p, ["start"])):
        with pytest.raises(ValueError, match="starts at byte .* of the file .* not a multiple of its"):
            reader(str(tmp_path / "odd_header.hshard"))
    # ... and the arrays a launch stages must be integer arrays that fit their staging type: refused on load, by name
    flat0 = shards.read_flat(path)
    for patch, message in ((dict(mapq0=np.asarray(flat0["mapq0"], np.float32)), "array mapq0 has dtype float32"),
                           (dict(cigars0=np.asarray(flat0["cigars0"], np.int64) + (1 << 33)), "array cigars0 .int64. holds values outside uint32"),
                           (dict(window_start=np.asarray(flat0["window_start"], np.float64)), "array window_start has dtype float64")):
        mistyped = shards.write_flat(str(tmp_path / "mistyped.hshard"), dict({k: np.array(v) for k, v in flat0.items()}, **patch))
        with pytest.raises(ValueError, match=message):
            shards.PackedShard.from_file(mistyped)
    wider = shards.write_flat(str(tmp_path / "wider.hshard"), dict({k: np.array(v) for k, v in flat0.items()}, mapq0=np.asarray(flat0["mapq0"], np.int64)))
    assert shards.PackedShard.from_file(wider).n_sites == shards.PackedShard.from_file(path).n_sites      # wider storage, values in range: fine
    # the load balancer's read totals honour has_second: stray second-technology counts of a single-technology shard weigh nothing
    from hello_amd import call as driver
    flat = shards.read_flat(path)
    assert int(np.asarray(flat["has_second"]).reshape(-1)[0]) == 1
    both = int(np.maximum(flat["reads_per_allele0"], 1).sum() + np.maximum(flat["reads_per_allele1"], 1).sum())
    first = int(np.maximum(flat["reads_per_allele0"], 1).sum())
    stray = shards.write_flat(str(tmp_path / "stray.hshard"), dict({k: np.array(v) for k, v in flat.items()}, has_second=np.array(0)))
    assert list(driver.shard_read_totals([path, stray], threads=1)) == [both, first]


def test_run_keeps_a_bounded_number_of_shards_in_memory(tmp_path):
    """ADVICE r02 / VERDICT r02 weak 2: whatever the number of shards, at most ``read_ahead`` loaded shards wait for the GPU
    and at most ``depth`` + 2 launches' shards wait for the record writer.  A stand-in scorer (random posteriors, no GPU)
    drives the real loop, reader pool, coalescing, record stage and file writing."""
    import gc
    import weakref
    rng = np.random.default_rng(2)
    template = shards._payload(random_sites(rng, 6))
    alive, peak = weakref.WeakSet(), [0]

    def loader(path):
        shard = shards.PackedShard(dict(template))
        shard.path = path
        alive.add(shard)
        peak[0] = max(peak[0], len(alive))
        return shard

    class Scorer:
        hybrid = uses_ref = False
        stage_seconds = 0.0

        def __init__(self):
            self.queue = []

        def submit(self, batch, tags):
            S = sum(sh.n_sites for sh in batch)
            aps = np.concatenate([sh.alleles_per_site for sh in batch])
            P = int((aps * (aps + 1) // 2).sum())
            self.queue.append(sp.Scored(list(batch), list(tags), rng.random((4, P)).astype(np.float32), None))
            gc.collect()
            return [self.queue.pop(0)] if len(self.queue) > 2 else []

        def flush(self):
            out, self.queue = self.queue, []
            return out

    n = 400
    stats = sp.run(None, [f"shard{k}.npz" for k in range(n)], lambda k: str(tmp_path / f"features{k}"), reader_threads=3,
                   record_threads=1, sites_per_launch=20, read_ahead=6, loader=loader, scorer=Scorer())
    assert stats.sites == 6 * n and len(stats.outputs) == n and stats.launches >= n // 4
    # 6 waiting + 2 launches of <= 4 shards in the scorer + 2 queued + 1 being written + the one in hand
    assert peak[0] <= 6 + 3 * 4 + 2 * 4 + 4 + 1, peak[0]
    assert all(sp.SENTINEL in open(tmp_path / f"features{k}.log").read() for k in (0, n - 1))
    total = sp.merge_final_vcf(stats.outputs, lambda names: "", str(tmp_path / "final.vcf"))
    assert total == sum(o.position.shape[0] for o in stats.outputs) > n


@pytest.mark.parametrize("hybrid,uses_ref", [(False, False), (True, True)])
def test_staging_block_of_a_coalesced_launch_equals_the_per_shard_walk(hybrid, uses_ref):
    """ShardScorer._fill lays a launch's shards out in the pinned block with ONE NumPy call per field (round 5: reference-sized
    shards spent more host time in per-shard call overhead than in copying).  Held here, without a GPU, to the plain per-shard
    walk: every field of every shard at its place, offsets and site indices shifted to the coalesced numbering -- ragged shards
    (one of a single site), one and two technologies, the one-hot reference segments."""
    rng = np.random.default_rng(11)
    payloads = [shards._payload(random_sites(rng, n, hybrid=hybrid)) for n in (5, 1, 9, 3)]
    packed = [shards.PackedShard(dict(p)) for p in payloads]
    if uses_ref:
        for sh in packed:
            sh.onehot = sh.segment_onehot()                            # what the reader threads attach (shard_pipeline.load_shard)
    scorer = sp.ShardScorer.__new__(sp.ShardScorer)                    # layout + fill need no engine
    scorer.hybrid, scorer.uses_ref, scorer.L = hybrid, uses_ref, 150
    techs = (0, 1) if hybrid else (0,)
    parts, nbytes, reads, S = scorer._layout(packed)
    block = np.full(nbytes, 0xAB, np.uint8)
    scorer._fill(packed, parts, block, techs)

    def view(key):
        at, dtype, count = parts[key]
        return block[at:at + count * dtype.itemsize].view(dtype)
    for t in techs:
        fa = [sh.featurizer_core(t) for sh in packed]
        for name in ("bases", "quals", "cigars"):
            want = np.concatenate([f[name] for f in fa] + [np.zeros(1, fa[0][name].dtype)])
            assert np.array_equal(view((name, t)), want), (name, t)
        for name, data in (("read_off", "bases"), ("cigar_off", "cigars")):
            want, shift = [np.zeros(1, np.int64)], 0
            for f in fa:
                want.append(f[name][1:] + shift)
                shift += int(f[data].shape[0])
            assert np.array_equal(view((name, t)), np.concatenate(want)), (name, t)
        for name in sp.PER_READ:
            want, site = [], 0
            for f, sh in zip(fa, packed):
                want.append(f[name] + site if name == "site_of_read" else f[name])
                site += sh.n_sites
            got = view((name, t))
            assert np.array_equal(got, np.concatenate(want).astype(got.dtype)), (name, t)
        assert reads[t] == sum(int(f["site_of_read"].shape[0]) for f in fa)
    assert np.array_equal(view(("ref", None)), np.concatenate([sh.ref for sh in packed] + [np.zeros(1, np.uint8)]))
    want, shift = [np.zeros(1, np.int64)], 0
    for sh in packed:
        want.append(sh.ref_off[1:] + shift)
        shift += int(sh.ref.shape[0])
    assert np.array_equal(view(("ref_off", None)), np.concatenate(want))
    for name, attr in (("window_start", "window_start"), ("asm_start", "start"), ("asm_stop", "stop")):
        assert np.array_equal(view((name, None)), np.concatenate([getattr(sh, attr) for sh in packed]))
    if uses_ref:
        assert np.array_equal(view(("onehot", None)), np.concatenate([sh.onehot.reshape(-1) for sh in packed]))
    assert S == sum(sh.n_sites for sh in packed)
