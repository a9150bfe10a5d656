"""How far inside the 1e-4 posterior tolerance the engine stays as the weights grow: the stress batches of
tests/test_gpu_parity.py (the five BASELINE configurations with ragged extremes: 1-read next to 1000-read alleles, dummy
reads only, identical alleles) at weight gains 0.5 ... 8 against the CPU oracle scored one site per call.  Nothing is
asserted: the table is the evidence (profiles/r03_parity_margin_{fp32,bf16x3}.txt; DESIGN.md section 4 quotes them).

    python tools/parity_margin.py [--gains 0.5,1,2.5,4,6,8] [--arithmetic fp32]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hello_amd import netspec as ns, synth, weights  # noqa: E402
from hello_amd.engine import Engine  # noqa: E402
from oracle import moe_oracle as mo  # noqa: E402
from tests.test_gpu_parity import BASELINE_CONFIGS, _with_extremes, sigmoid  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gains", default="0.5,1,2.5,4,6,8")
    ap.add_argument("--arithmetic", default="fp32")
    args = ap.parse_args()
    gains = [float(g) for g in args.gains.split(",")]
    print(f"# python tools/parity_margin.py --gains {args.gains} --arithmetic {args.arithmetic}")
    print(f"{'configuration':46s} {'gain':>5s} {'max |logit|':>12s} {'|dlogit|/scale':>15s} {'|d allele prob|':>16s} {'|d posterior|':>14s}")
    worst = {}
    for label, cfg, kw in BASELINE_CONFIGS:
        spec = ns.build(cfg)
        for gain in gains:
            state = weights.synth_state(spec, seed=77, gain=gain)
            hybrid = "hybrid_coverage" in kw
            batch = _with_extremes(synth.make_sites(36, seed=int(1000 * gain) + len(cfg), **kw), 4000 + int(10 * gain), hybrid,
                                   kw.get("channels", 6))
            eng = Engine(spec, state, device=0, arithmetic=args.arithmetic)
            logits, meta, post = eng.forward_batch(batch, posteriors=True)
            eng.close()
            want, want_meta = mo.forward_batch(mo.Oracle(spec, state, backend="torch"), batch, chunk_sites=1)
            scale = max(1.0, float(np.abs(want).max()))
            aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
            col, dpost = 0, 0.0
            for s in range(batch.n_sites):
                probs = [mo.sigmoid(want[e, aoff[s]:aoff[s + 1]]) for e in range(want.shape[0])]
                if len(probs) == 1:
                    probs += [np.zeros_like(probs[0])] * 2
                m = want_meta[s] if want_meta is not None else np.array([1, 0, 0], np.float32)
                rows = mo.posteriors(probs, m)
                n = rows[0].shape[0]
                dpost = max(dpost, max(float(np.abs(post[r, col:col + n] - rows[r]).max()) for r in range(4)))
                col += n
            dprob = float(np.abs(sigmoid(logits) - sigmoid(want)).max())
            worst[gain] = max(worst.get(gain, 0.0), dpost)
            print(f"{label[:46]:46s} {gain:5.1f} {scale:12.4g} {np.abs(logits - want).max() / scale:15.2e} {dprob:16.2e} {dpost:14.2e}")
    print("worst posterior difference per gain: " + ", ".join(f"{g:g}: {worst[g]:.2e}" for g in gains))


if __name__ == "__main__":
    main()
