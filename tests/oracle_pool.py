"""The CPU oracle over EVERY site of a full launch, as a separate process tree (test infrastructure).

A GPU test starts this module as a child (``python -m tests.oracle_pool ...``) before it launches the engine, so the oracle's
worker pool is forked by a process that never touches the GPU (fork after HIP initialisation is not safe) and the CPU answers
are computed while the GPU scores the same sites.  The batch is regenerated here from its seed (hello_amd.synth is
deterministic), every site goes through oracle/moe_oracle.py ONE SITE PER CALL -- the reference's per-site form,
MixtureOfExpertsAdvanced.py:520-589 on top of :161-252 -- on a pool of single-threaded workers, and the answers are written to
an .npz: ``probs`` [E, A] = sigmoid(logit) per expert, ``meta`` [S, 3] (absent for single-expert models), ``post`` [4, P] pair
posteriors (rows mix, e0, e1, e2; pairs per site in first-seen itertools.product order).

    python -m tests.oracle_pool --spec hybrid_full --spec-kw '{}' --weights-seed 33 --sites 8192 --sites-seed 711
                                --sites-kw '{"coverage": 30, "hybrid_coverage": 15}' --out answers.npz
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_JOB = {}


def _tuplify(kw):
    return {k: tuple(v) if isinstance(v, list) else v for k, v in kw.items()}


def _worker(span):
    lo, hi = span
    import torch
    torch.set_num_threads(1)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import moe_oracle as mo
    oracle, batch = _JOB["oracle"], _JOB["batch"]
    sub = batch.site_slice(lo, hi)
    logits, meta = mo.forward_batch(oracle, sub, chunk_sites=1)
    probs = mo.sigmoid(logits)
    aoff = np.concatenate([[0], np.cumsum(sub.alleles_per_site)])
    post = []
    for s in range(sub.n_sites):
        p = [probs[e, aoff[s]:aoff[s + 1]] for e in range(probs.shape[0])]
        if len(p) == 1:                     # single-expert models: experts [e0, 0, 0], meta [1, 0, 0]  (:530-538)
            p, m = p + [np.zeros_like(p[0])] * 2, np.array([1, 0, 0], np.float32)
        else:
            m = meta[s]
        post.append(np.stack(mo.posteriors(p, m)))
    return lo, logits, probs, meta, np.concatenate(post, axis=1)


def run(spec_name, spec_kw, weights_seed, n_sites, sites_seed, sites_kw, backend="torch", workers=None):
    """-> dict(logits, probs, meta | None, post, seconds, workers).  Forks: call from a process that has not touched the GPU."""
    import multiprocessing as mp
    import torch  # noqa: F401  -- imported once here (CPU only), inherited by the forked workers
    import bench
    from hello_amd import netspec as ns, synth, weights
    from oracle import moe_oracle as mo
    spec = ns.build(spec_name, **spec_kw)
    _JOB["oracle"] = mo.Oracle(spec, weights.synth_state(spec, seed=weights_seed), backend=backend)
    _JOB["batch"] = synth.make_sites(n_sites, seed=sites_seed, **_tuplify(sites_kw))
    workers = workers or bench.host_cores()
    step = max(1, min(64, -(-n_sites // (4 * workers))))
    spans = [(lo, min(lo + step, n_sites)) for lo in range(0, n_sites, step)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        parts = sorted(pool.map(_worker, spans, chunksize=1), key=lambda p: p[0])
    dt = time.perf_counter() - t0
    meta = None if parts[0][3] is None else np.concatenate([p[3] for p in parts], axis=0)
    return dict(logits=np.concatenate([p[1] for p in parts], axis=1), probs=np.concatenate([p[2] for p in parts], axis=1), meta=meta,
                post=np.concatenate([p[4] for p in parts], axis=1), seconds=dt, workers=workers)


def start(spec_name, weights_seed, n_sites, sites_seed, sites_kw, out, spec_kw=None, backend="torch"):
    """Start the oracle over a whole batch as a child process (it forks its workers; this process may hold the GPU).  -> Popen."""
    import subprocess
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    return subprocess.Popen([sys.executable, "-m", "tests.oracle_pool", "--spec", spec_name, "--spec-kw", json.dumps(spec_kw or {}),
                             "--weights-seed", str(weights_seed), "--sites", str(n_sites), "--sites-seed", str(sites_seed),
                             "--sites-kw", json.dumps(sites_kw), "--backend", backend, "--out", out], cwd=ROOT, env=env)


def collect(proc, out, timeout=900):
    """Wait for a ``start``ed oracle and load its answers."""
    rc = proc.wait(timeout=timeout)
    if rc != 0:
        raise RuntimeError(f"the oracle pool exited with status {rc}")
    with np.load(out) as z:
        got = {k: z[k] for k in z.files}
    got["meta"] = got["meta"] if got["meta"].size else None
    got["seconds"], got["workers"] = float(got["seconds"]), int(got["workers"])
    return got


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", required=True)
    ap.add_argument("--spec-kw", default="{}")
    ap.add_argument("--weights-seed", type=int, required=True)
    ap.add_argument("--sites", type=int, required=True)
    ap.add_argument("--sites-seed", type=int, required=True)
    ap.add_argument("--sites-kw", default="{}")
    ap.add_argument("--backend", default="torch", choices=["torch", "numpy"])
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    got = run(a.spec, json.loads(a.spec_kw), a.weights_seed, a.sites, a.sites_seed, json.loads(a.sites_kw), a.backend, a.workers or None)
    np.savez(a.out, logits=got["logits"], probs=got["probs"], meta=got["meta"] if got["meta"] is not None else np.zeros((0, 3), np.float32),
             post=got["post"], seconds=got["seconds"], workers=got["workers"])


if __name__ == "__main__":
    main()
