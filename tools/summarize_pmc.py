"""Summarise a rocprofv3 counter_collection.csv per kernel (mean counter value per dispatch), or -- with --traffic DIR --
derive profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE / SQ summaries in DIR (gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 128-B requests at 64 B -> doubled; both in KB)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"hello::(\w+)", name)
    return m.group(1) if m else name[:60]


def summarise(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as fh:
        for row in csv.DictReader(fh):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, counters in acc.items():
        if not k.startswith(("readconv", "conv1d", "segsum", "mix", "head", "posteriors")):
            continue
        out[k] = {c: {"mean": sum(v) / len(v), "sum": sum(v), "dispatches": len(v)} for c, v in counters.items()}
    return out


def main():
    if sys.argv[1] == "--traffic":
        d = sys.argv[2]
        load = lambda name: json.load(open(os.path.join(d, next(f for f in os.listdir(d) if f.endswith(f"pmc_{name}.txt")))))   # noqa: E731
        # readconv_kernel is launched twice per forward (bulk + remainder, readconv_plan): per-forward figures are the
        # kernel's SUM over the pass divided by the number of forwards (= dispatches of its finalize kernel)
        def per_forward(name, counter):
            t = load(name)
            return t["readconv_kernel"][counter]["sum"] / t["readconv_finalize_kernel"][counter]["dispatches"]
        fetch = per_forward("FETCH_SIZE", "FETCH_SIZE")
        write = per_forward("WRITE_SIZE", "WRITE_SIZE")
        sq = load("SQ_VALU_MFMA_BUSY_CYCLES")["readconv_kernel"]
        g = lambda c: sq[c]["sum"]                                                                                                # noqa: E731
        print(json.dumps({
            "kernel": "hello::readconv_kernel (bench.py headline loop, 8 192 sites / 246 k reads per forward; two launches per "
                      "forward: 7 680 workgroups of 8 groups, then 60 of one group)",
            "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
            "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 (FETCH_SIZE tallies 128-B requests at 64 B); separate --pmc passes",
            "bytes_per_launch": (2 * fetch + write) * 1024,
            "algorithmic_bytes_per_launch": 246002 * 900 + 17646 * 36 * 64 * 4,
            # 1024 SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs: busy cycles per SIMD / active cycles
            "sq": {"SQ_VALU_MFMA_BUSY_CYCLES_over_GRBM_GUI_ACTIVE_over_128": g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") * 128),
                   "SQ_WAIT_ANY_over_SQ_WAVE_CYCLES": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
                   "SQ_LDS_BANK_CONFLICT_over_SQ_LDS_IDX_ACTIVE": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")}}, indent=1))
        return
    print(json.dumps(summarise(sys.argv[1]), indent=1))


if __name__ == "__main__":
    main()
