"""Shared helpers for the test-suite (fixture loading, state digests)."""
import hashlib
import os

import numpy as np

from hello_amd import netspec as ns
from hello_amd import synth, weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["single_tech_batched", "single_tech_bn", "single_tech_hp", "single_tech_deep",
            "hybrid_no_ensemble", "hybrid_full", "hybrid_ensemble2", "hybrid_compressor2", "merged_single", "merged_hybrid", "single_tech_addendum", "hybrid_no_ensemble_addendum", "single_tech_softplus", "hybrid_no_ensemble_wide",
            "merged_hybrid_250", "single_tech_layernorm"]


def state_digest(state):
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(np.ascontiguousarray(state[k]).tobytes())
    return h.hexdigest()[:16]


def load_fixture(name):
    """-> (spec, state, SiteBatch, dict of expected arrays)"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg, norm = str(z["config"]), str(z["norm"])
    spec = ns.build(cfg, norm=norm) if norm != "wn" else ns.build(cfg)
    state = weights.synth_state(spec, seed=int(z["weight_seed"]))
    assert state_digest(state) == str(z["state_digest"]), (
        "synthetic weight generator drifted from the one the golden vectors were made with; "
        "re-run tests/golden/make_fixtures.py in the build container")
    batch = synth.SiteBatch(
        z["reads0"], z["reads_per_allele0"], z["alleles_per_site"], z["ref_onehot"],
        z["reads1"] if "reads1" in z.files else None,
        z["reads_per_allele1"] if "reads_per_allele1" in z.files else None)
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    return spec, state, batch, exp


def canonical_vcf_line(line):
    """A record line with its ALT alleles in sorted order and the genotype indices remapped accordingly: the
    reference orders ALTs by list(set(...)) (per-process hash order, prepareVcf.py:63,75), so two runs are compared
    in this form.  None stays None."""
    if line is None:
        return None
    f = line.split("\t")
    alts = f[4].split(",")
    order = sorted(range(len(alts)), key=lambda i: alts[i])
    new_index = {old + 1: new + 1 for new, old in enumerate(order)}
    new_index[0] = 0
    f[4] = ",".join(alts[i] for i in order)
    f[9] = "/".join(str(new_index[int(g)]) for g in f[9].split("/"))
    return "\t".join(f)


def load_vcf_reference():
    """tests/golden/vcf_reference.json with pair keys back as tuples."""
    import json
    z = json.load(open(os.path.join(GOLDEN, "vcf_reference.json")))
    unkey = lambda d: {tuple(k.split("|")): v for k, v in d.items()}           # noqa: E731
    for c in z["calls"]:
        c["likelihoods"] = unkey(c["likelihoods"])
    for case in z["caller"]:
        for site in case["sites"]:
            if site["features"] is not None:
                site["features"]["expertPredictions"] = [unkey(e) for e in site["features"]["expertPredictions"]]
    for item in z["shard"]["items"]:
        item["expertPredictions"] = tuple(unkey(e) for e in item["expertPredictions"])
    return z


def caller_case_sites(case):
    """(spec, state, [(featureDict of uint8 arrays, ref one-hot [1, L, 5] uint8, fixture site)]) of one caller case of
    vcf_reference.json: the pileups are regenerated from the stored seed (checked against the stored digest)."""
    z = load_vcf_reference() if isinstance(case, int) else None
    case = z["caller"][case] if z is not None else case
    spec = ns.build(case["config"])
    state = weights.synth_state(spec, seed=case["weight_seed"])
    batch = synth.make_sites(len(case["sites"]), seed=case["batch_seed"], **case["batch_kwargs"])
    assert hashlib.sha256(batch.reads0.tobytes()).hexdigest()[:16] == case["reads0_sha"], "synthetic generator drifted"
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    r0 = np.concatenate([[0], np.cumsum(batch.reads_per_allele0)])
    r1 = None if batch.reads1 is None else np.concatenate([[0], np.cumsum(batch.reads_per_allele1)])
    out = []
    for s, site in enumerate(case["sites"]):
        fd = {}
        for name, a in zip(site["alleles"], range(aoff[s], aoff[s + 1])):
            fd[name] = (batch.reads0[r0[a]:r0[a + 1]], None if r1 is None else batch.reads1[r1[a]:r1[a + 1]])
        out.append((fd, site))
    return spec, state, out


def reference_segment_onehot(genome, start, stop, span=150):
    """caller_calling.py:53-97 (one_hot_encode / get_reference_segment): uint8 [1, span, 5], order ACGT + other."""
    mid = (start + stop) // 2
    left = mid - span // 2
    seg = genome[left:left + span]
    out = np.zeros((1, len(seg), 5), np.uint8)
    out[0, np.arange(len(seg)), ["ACGT".find(b) if b in "ACGT" else 4 for b in seg]] = 1
    return out


def canonical_pickle(name, tmp_dir):
    """tests/golden/canonical_<name>.wrapper.dnn.gz (a canonical-size reference pickle with zeroed parameters, written by the
    reference's own torch.save in make_fixtures.py) unpacked into ``tmp_dir`` -> path of the .wrapper.dnn file."""
    import gzip
    out = os.path.join(str(tmp_dir), f"canonical_{name}.wrapper.dnn")
    with gzip.open(os.path.join(GOLDEN, f"canonical_{name}.wrapper.dnn.gz"), "rb") as src, open(out, "wb") as dst:
        dst.write(src.read())
    return out


def oracle_per_site(spec, state, batch, backend="torch", workers=8):
    """``mo.forward_batch(Oracle(spec, state, backend), batch, chunk_sites=1)`` -- the oracle one site per call, the reference's per-site
    form -- with the sites spread over ``workers`` threads (the convolution back ends release the interpreter lock; every thread
    its own Oracle; one BLAS / torch thread each, so that a box with few or slow cores is not oversubscribed).  Same numbers: a
    site's result does not depend on the others'.  -> (logits [E, A], meta [S, 3] | None)."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from oracle import moe_oracle as mo
    n = batch.n_sites
    workers = max(1, min(workers, n, len(os.sched_getaffinity(0))))
    edges = np.linspace(0, n, workers + 1).astype(int)
    spans = [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]
    before = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with ThreadPoolExecutor(len(spans)) as pool:
            parts = list(pool.map(lambda ab: mo.forward_batch(mo.Oracle(spec, state, backend=backend), batch.site_slice(*ab), chunk_sites=1), spans))
    finally:
        torch.set_num_threads(before)
    logits = np.concatenate([p[0] for p in parts], axis=1)
    meta = None if parts[0][1] is None else np.concatenate([p[1] for p in parts], axis=0)
    return logits, meta
