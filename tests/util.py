"""Shared helpers for the test-suite (fixture loading, state digests)."""
import hashlib
import os

import numpy as np

from hello_amd import netspec as ns
from hello_amd import synth, weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["single_tech_batched", "single_tech_bn", "single_tech_hp", "single_tech_deep",
            "hybrid_no_ensemble", "hybrid_full", "hybrid_ensemble2", "merged_single", "merged_hybrid", "single_tech_addendum", "hybrid_no_ensemble_addendum", "single_tech_softplus", "hybrid_no_ensemble_wide",
            "merged_hybrid_250"]


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
