"""Summaries of rocprofv3 output for profiles/ (every kernel of the run, no name list):

    summarize_pmc.py COUNTER_COLLECTION.csv            per kernel and counter: mean / sum / dispatches (JSON)
    summarize_pmc.py --traffic DIR                     profiles/hbm_traffic.json for the dominant kernel from the
                                                       FETCH_SIZE / WRITE_SIZE / SQ summaries in DIR
    summarize_pmc.py --table DIR KERNEL_STATS.csv N    Markdown table, one row per kernel of the per-forward launch list:
                                                       launches and time per forward (N forwards in the stats run), share of
                                                       the forward, matrix-pipe busy share, wave wait share, LDS bank-conflict
                                                       share, HBM bytes per forward

gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 128-B requests at 64 B ->
doubled; both counters in KB.  SQ ratios: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) (1 024 SIMDs, GRBM_GUI_ACTIVE
summed over the 8 XCDs), SQ_WAIT_ANY / SQ_WAVE_CYCLES, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    """hello::kernel<template args> -> kernel (template arguments dropped: instantiations of one kernel are one row)."""
    m = re.search(r"hello::(?:\w+::)*(\w+)", name)
    return m.group(1) if m else re.sub(r"\(.*", "", name)[:60]


def summarise(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as fh:
        for row in csv.DictReader(fh):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: {"mean": sum(v) / len(v), "sum": sum(v), "dispatches": len(v)} for c, v in counters.items()}
            for k, counters in acc.items()}


def load(d, name):
    path = next((os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(f"pmc_{name}.txt")), None)
    return json.load(open(path)) if path else {}


def table(d, stats_csv, forwards):
    rows = defaultdict(lambda: dict(calls=0, ns=0.0))
    with open(stats_csv) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Name"])
            rows[k]["calls"] += int(r["Calls"])
            rows[k]["ns"] += float(r["TotalDurationNs"])
    sq, fetch, write = load(d, "SQ_VALU_MFMA_BUSY_CYCLES"), load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    total = sum(v["ns"] for v in rows.values())
    out = ["| kernel | launches / forward | ms / forward | share | matrix pipe busy | waves waiting | LDS bank conflicts | HBM MB / forward (fetch + write) |",
           "|---|---|---|---|---|---|---|---|"]

    def ratio(t, k, num, den, scale=1.0):
        c = t.get(k, {})
        if num in c and den in c and c[den]["sum"] > 0:
            return f"{100.0 * c[num]['sum'] / (c[den]['sum'] * scale):.1f} %"
        return "n/a"

    def mb(k):
        f, w = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
        if not f or not w:
            return "n/a"
        per_call = lambda c: c["sum"] / c["dispatches"]                       # noqa: E731
        calls_per_forward = rows[k]["calls"] / forwards
        return f"{(2 * per_call(f)) * 1024 * calls_per_forward / 1e6:.1f} + {per_call(w) * 1024 * calls_per_forward / 1e6:.1f}"
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ns"]):
        if not re.match(r"[a-z_0-9]+_kernel$", k):
            continue                                                          # torch's own fill / copy kernels
        out.append(f"| `{k}` | {v['calls'] / forwards:g} | {v['ns'] / forwards / 1e6:.3f} | {100 * v['ns'] / total:.1f} % | "
                   f"{ratio(sq, k, 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 128.0)} | {ratio(sq, k, 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')} | "
                   f"{ratio(sq, k, 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')} | {mb(k)} |")
    return "\n".join(out)


def provenance():
    """Which binary the passes profiled: sha256 of the library the profiled bench.py LOADED -- HELLO_LIB when it is set (the path
    hello_amd.engine opens, read the same way here without importing torch), else the in-tree hello_amd/libhello_mi355x.so -- and
    the commit the caller names ($HELLO_PROFILE_COMMIT: the GPU box's copy of the tree has no .git).  bench.py reports
    `roofline.traffic` only when this hash equals the hash of the library it loaded."""
    import hashlib
    lib = os.environ.get("HELLO_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hello_amd",
                                                      "libhello_mi355x.so")
    try:
        sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    except OSError:
        sha = None
    return {"lib_sha256": sha, "lib_path": lib, "commit": os.environ.get("HELLO_PROFILE_COMMIT") or None}


def traffic(d):
    # readconv_kernel is launched twice per forward (bulk + remainder, readconv_plan): per-forward figures are the
    # kernel's SUM over the pass divided by the number of forwards (= dispatches of its finalize kernel)
    def per_forward(name, counter):
        t = load(d, name)
        return t["readconv_kernel"][counter]["sum"] / t["readconv_finalize_kernel"][counter]["dispatches"]
    fetch = per_forward("FETCH_SIZE", "FETCH_SIZE")
    write = per_forward("WRITE_SIZE", "WRITE_SIZE")
    sq = load(d, "SQ_VALU_MFMA_BUSY_CYCLES")["readconv_kernel"]
    g = lambda c: sq[c]["sum"]                                                                                                # noqa: E731
    # the WHOLE forward: every kernel of the pass (the engine's own and torch's copy kernels), per forward
    ft, wt = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    forwards = ft["readconv_finalize_kernel"]["FETCH_SIZE"]["dispatches"]
    by_kernel = {}
    for k in sorted(set(ft) | set(wt)):
        f_kb = ft.get(k, {}).get("FETCH_SIZE", {}).get("sum", 0.0) / forwards
        w_kb = wt.get(k, {}).get("WRITE_SIZE", {}).get("sum", 0.0) / wt["readconv_finalize_kernel"]["WRITE_SIZE"]["dispatches"]
        by_kernel[k] = round((2 * f_kb + w_kb) * 1024)
    return {
        **provenance(),
        "bytes_per_forward": sum(by_kernel.values()),
        "algorithmic_bytes_per_forward": 246002 * 900 + 4 * (17646 + 8193) + 4 * 17646 + 16 * 26400,
        "algorithmic_bytes_per_forward_note": "8 192-site launch: 900 B per read + CSR counts in, 4 B per allele (logits) + 16 B per pair (posteriors) out",
        "bytes_per_forward_by_kernel": dict(sorted(by_kernel.items(), key=lambda kv: -kv[1])),
        "kernel": "hello::readconv_kernel (bench.py headline loop, 8 192 sites / 246 k reads per forward; two launches per "
                  "forward: 7 680 workgroups of 8 groups, then 60 of one group)",
        "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
        "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 (FETCH_SIZE tallies 128-B requests at 64 B); separate --pmc passes",
        "bytes_per_launch": (2 * fetch + write) * 1024,
        "algorithmic_bytes_per_launch": 246002 * 900 + 17646 * 36 * 64 * 4,
        "sq": {"SQ_VALU_MFMA_BUSY_CYCLES_over_GRBM_GUI_ACTIVE_over_128": g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") * 128),
               "SQ_WAIT_ANY_over_SQ_WAVE_CYCLES": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
               "SQ_LDS_BANK_CONFLICT_over_SQ_LDS_IDX_ACTIVE": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")}}


def main():
    if sys.argv[1] == "--traffic":
        print(json.dumps(traffic(sys.argv[2]), indent=1))
    elif sys.argv[1] == "--table":
        print(table(sys.argv[2], sys.argv[3], float(sys.argv[4])))
    else:
        print(json.dumps(summarise(sys.argv[1]), indent=1))


if __name__ == "__main__":
    main()
