"""Where readconv_kernel's time goes, section by section, MEASURED.

The kernel's 20 barriers per group of 4 reads cut its instruction stream into 21 sections (one layer each: stem, six 32-channel
convolutions, the strided block, seven 64-channel convolutions, the per-allele sum).  A stamped instantiation of the very same
kernel (hello_engine_debug_stamps: s_memtime per wave at the start of each group and on both sides of every barrier, written to
memory nothing reads) runs one 8 192-site forward with TWO workgroups per CU (the product's occupancy: two waves per SIMD, one of
each workgroup) and one with ONE workgroup per CU (LDS padding).  Per section and wave this gives

    compute = arrival at the section's closing barrier - release from the previous one      wait = release - arrival

and, from the product kernel's assembly, the section's instruction counts by class.  Printed:

  1. the table: static counts | cycles per wave and group, 1 and 2 workgroups per CU, compute and barrier wait
  2. the sum of the rows x groups per workgroup slot / the in-kernel clock against the launch's HIP-event time (must agree)
  3. a MEASURED cost model: non-negative least squares of the sections' SIMD time (1 WG/CU: the wave's own; 2 WG/CU: half of the
     pair's) on their instruction counts -> cycles per MFMA, per plain VALU, per packed VALU, per DPP VALU, per LDS read, per LDS
     write, per VMEM, per barrier -- and its residual per section
  4. the matrix pipe's idle share split into {barrier wait, own stalls (LDS / memory latency, MFMA dependency, waitcnt),
     vector + LDS + scalar issue, arbitration between the two waves of a SIMD}

    python tools/readconv_section_stamps.py [--sites 8192] [--static tools/_bin/readconv_sections.json]
    python tools/readconv_section_stamps.py --static-only   # (no GPU) writes the static table the GPU run reads
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

SYMBOL = "_ZN5hello15readconv_kernelINS_2rc3CfgILi4ELi4ELi150ELi0EEELb1ELi3ELb1ELb0ELb0ELb0EEEvNS_12ReadConvArgsE"
CLASSES = ["mfma", "valu", "valu_pk", "valu_dpp", "lds_read", "lds_write", "vmem", "salu", "waitcnt"]
NAMES = ["prologue: weights + bytes -> LDS", "stem conv1 (bytes -> 16)", "stem conv2 F(2,3) 16 -> 16", "stem conv3 + pool F(2,3) 16 -> 32",
         "zero rows", "RB32.0 conv a F(3,3)", "RB32.0 conv b", "RB32.1 conv a", "RB32.1 conv b", "RB32.2 conv a", "RB32.2 conv b",
         "strided 32 -> 64 + 1x1 shortcut (direct)", "strided block conv 2 F(3,3) 64 -> 64", "RB64.0 conv a", "RB64.0 conv b", "RB64.1 conv a",
         "RB64.1 conv b", "RB64.2 conv a", "RB64.2 conv b", "per-allele sum", "flush (per workgroup)"]


def static_sections():
    import check_asm_hazards as cah
    lines = cah.assembly(os.path.join(ROOT, "hello_amd", "csrc", "readconv_fused.hip"))
    inside, out = False, []
    cur = dict.fromkeys(CLASSES, 0)
    for ln in lines:
        t = ln.strip()
        if not inside:
            inside = t.startswith(SYMBOL + ":")
            continue
        op = t.split()[0] if t and t[0].isalpha() and not t.endswith(":") else ""
        if not op:
            continue
        if op.startswith("v_mfma"):
            kind = "mfma"
        elif op.startswith("v_pk_"):
            kind = "valu_pk"
        elif op.startswith("v_") and ("row_shl" in t or "row_shr" in t or "quad_perm" in t or "row_ror" in t or "_dpp" in op):
            kind = "valu_dpp"
        elif op.startswith("v_"):
            kind = "valu"
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            kind = "lds_read"
        elif op.startswith("ds_"):
            kind = "lds_write"
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            kind = "vmem"
        elif op == "s_waitcnt":
            kind = "waitcnt"
        else:
            kind = "salu"
        cur[kind] += 1
        if op in ("s_barrier", "s_endpgm"):
            out.append(cur)
            cur = dict.fromkeys(CLASSES, 0)
        if op == "s_endpgm":
            break
    return out


def in_kernel(stamps, bulk):
    """stamps [wgs, 4, groups, 48] -> per (workgroup of the bulk launch, wave, group) compute[20], wait[20], group cycles, clock."""
    s = stamps[:bulk].astype(np.int64)
    ok = (s[..., 0] > 0) & (s[..., 40] > 0)
    arrive, release = s[..., 1:41:2], s[..., 2:42:2]
    prev = np.concatenate([s[..., 0:1], release[..., :-1]], axis=-1)
    compute, wait = (arrive - prev)[ok], (release - arrive)[ok]
    total = (release[..., 19] - s[..., 0])[ok]
    ticks, real = (s[..., 40] - s[..., 0])[ok].sum(), (s[..., 43] - s[..., 42])[ok].sum()
    clock_ghz = ticks / max(real, 1) * 0.1
    # the group loop's own cost between groups (release of barrier 19 -> next group's start) and the final flush
    nxt = s[:, :, 1:, 0] - s[:, :, :-1, 40]
    between = nxt[(s[:, :, 1:, 0] > 0) & (s[:, :, :-1, 40] > 0)]
    last = np.where(ok, np.arange(s.shape[2])[None, None, :], -1).max(axis=2)
    flush = []
    for w in range(0, s.shape[0], max(1, s.shape[0] // 512)):
        for v in range(s.shape[1]):
            if last[w, v] >= 0 and s[w, v, 0, 44] > 0:
                flush.append(s[w, v, 0, 44] - s[w, v, last[w, v], 40])
    return dict(compute=compute.mean(axis=0), wait=wait.mean(axis=0), total=float(total.mean()), clock_ghz=float(clock_ghz),
                between=float(between.mean()) if between.size else 0.0, flush=float(np.mean(flush)) if flush else 0.0,
                n=int(ok.sum()), compute_p10=np.percentile(compute, 10, axis=0), compute_p90=np.percentile(compute, 90, axis=0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=8192)
    ap.add_argument("--static", default=os.path.join(ROOT, "tools", "_bin", "readconv_sections.json"))
    ap.add_argument("--static-only", action="store_true")
    args = ap.parse_args()
    if args.static_only or not os.path.exists(args.static):
        secs = static_sections()
        os.makedirs(os.path.dirname(args.static), exist_ok=True)
        json.dump(secs, open(args.static, "w"))
        if args.static_only:
            for i, c in enumerate(secs):
                print(i, c)
            return
    secs = json.load(open(args.static))
    assert len(secs) == 21, len(secs)

    import torch
    from hello_amd import netspec as ns, synth, weights
    from hello_amd.engine import Engine
    spec = ns.build("single_tech")
    eng = Engine(spec, weights.synth_state(spec, seed=1), device=0, arithmetic="fp32")
    batch = synth.make_sites(args.sites, seed=1001, coverage=30)
    reads = torch.from_numpy(batch.reads0).cuda()
    n_reads = int(batch.reads0.shape[0])

    def forward_ms(reps):
        eng.set_profiling(reps, only="readconv_fused")
        for _ in range(reps):
            eng.forward(reads, batch.reads_per_allele0, batch.alleles_per_site, posteriors=True)
        torch.cuda.synchronize()
        rows, n = eng.op_times_ms()
        eng.set_profiling(0)
        return max(r[2] for r in rows)
    for _ in range(3):
        eng.forward(reads, batch.reads_per_allele0, batch.alleles_per_site, posteriors=True)
    plain_ms = forward_ms(10)
    runs = {}
    for label, mode in (("2 workgroups per CU", 1), ("1 workgroup per CU", 3)):
        eng.record_stamps(mode)
        eng.forward(reads, batch.reads_per_allele0, batch.alleles_per_site, posteriors=True)     # warm (first launch of the instantiation)
        ms = forward_ms(4)
        stamps, bulk = eng.read_stamps()
        runs[label] = dict(in_kernel(stamps, bulk), ms=ms, wgs=int(stamps.shape[0]), bulk=bulk, groups_per_wg=int(stamps.shape[2]))
    eng.record_stamps(0)
    again_ms = forward_ms(10)
    eng.close()

    two, one = runs["2 workgroups per CU"], runs["1 workgroup per CU"]
    print(f"# tools/readconv_section_stamps.py: {args.sites} sites, {n_reads} reads, {(n_reads + 3) // 4} groups of 4 reads; readconv_kernel + finalize per forward (HIP events):")
    print(f"#   product kernel {plain_ms:.3f} ms (again after the stamped runs: {again_ms:.3f});  stamped, 2 workgroups per CU {two['ms']:.3f} ms;  stamped, 1 workgroup per CU {one['ms']:.3f} ms")
    print(f"#   in-kernel clock (s_memtime / s_memrealtime): {two['clock_ghz']:.3f} GHz at 2 workgroups per CU, {one['clock_ghz']:.3f} GHz at 1;  "
          f"{two['n']} / {one['n']} (workgroup, wave, group) records; bulk launch {two['bulk']} workgroups x {two['groups_per_wg']} groups")
    print("#   cycles are s_memtime ticks per WAVE and GROUP (mean over all records); est = 32 MFMA + 4 (VALU + LDS + VMEM) + SALU (the round-1 issue model)")
    hdr = (f"{'section':44s} {'MFMA':>5s} {'VALU':>5s} {'pk':>4s} {'dpp':>4s} {'LDSr':>5s} {'LDSw':>5s} {'VMEM':>5s} {'SALU':>5s} | {'est':>7s} | "
           f"{'1WG comp':>9s} {'1WG wait':>9s} | {'2WG comp':>9s} {'2WG wait':>9s} {'2WG p10-p90 comp':>17s} | {'(2WG c+w)/2':>11s} {'/est':>5s}")
    print(hdr)
    tot = dict.fromkeys(CLASSES, 0)
    est_tot = 0
    for i in range(20):
        c = secs[i]
        for k in CLASSES:
            tot[k] += c[k]
        est = 32 * c["mfma"] + 4 * (c["valu"] + c["valu_pk"] + c["valu_dpp"] + c["lds_read"] + c["lds_write"] + c["vmem"]) + c["salu"] + c["waitcnt"]
        est_tot += est
        half = (two["compute"][i] + two["wait"][i]) / 2
        print(f"{i:2d} {NAMES[i]:41s} {c['mfma']:5d} {c['valu']:5d} {c['valu_pk']:4d} {c['valu_dpp']:4d} {c['lds_read']:5d} {c['lds_write']:5d} {c['vmem']:5d} "
              f"{c['salu'] + c['waitcnt']:5d} | {est:7d} | {one['compute'][i]:9.0f} {one['wait'][i]:9.0f} | {two['compute'][i]:9.0f} {two['wait'][i]:9.0f} "
              f"{two['compute_p10'][i]:8.0f}-{two['compute_p90'][i]:<8.0f} | {half:11.0f} {half / max(est, 1):5.2f}")
    half_tot = (two["compute"].sum() + two["wait"].sum()) / 2
    print(f"{'sum of the 20 sections':44s} {tot['mfma']:5d} {tot['valu']:5d} {tot['valu_pk']:4d} {tot['valu_dpp']:4d} {tot['lds_read']:5d} {tot['lds_write']:5d} "
          f"{tot['vmem']:5d} {tot['salu'] + tot['waitcnt']:5d} | {est_tot:7d} | {one['compute'].sum():9.0f} {one['wait'].sum():9.0f} | "
          f"{two['compute'].sum():9.0f} {two['wait'].sum():9.0f} {'':17s} | {half_tot:11.0f} {half_tot / est_tot:5.2f}")
    print(f"   between groups (loop back edge) {one['between']:.0f} / {two['between']:.0f} cycles; final flush per workgroup {one['flush']:.0f} / {two['flush']:.0f} (1 / 2 workgroups per CU)")

    # ---- 2. do the rows add up to the launch? ----------------------------------------------------------------------------
    print("\n# Do the rows add up to the kernel's time?  A workgroup slot walks its groups back to back; a launch = rounds x groups per workgroup x")
    print("# (sum of the sections + back edge) / clock, + the second launch's one-group workgroups (one more group time) + launch gaps.")
    groups = (n_reads + 3) // 4
    for label, r, slots in (("2 workgroups per CU", two, 512), ("1 workgroup per CU", one, 256)):
        per_group = r["compute"].sum() + r["wait"].sum() + r["between"]
        rounds = r["bulk"] / slots
        rest_rounds = -(-(r["wgs"] - r["bulk"]) // slots) if r["wgs"] > r["bulk"] else 0
        modeled = (rounds * r["groups_per_wg"] + rest_rounds) * per_group / (r["clock_ghz"] * 1e6)
        print(f"#   {label}: {per_group:.0f} cycles per group and slot x ({rounds:.2f} rounds x {r['groups_per_wg']} groups + {rest_rounds} round of one-group workgroups) "
              f"/ {r['clock_ghz']:.3f} GHz = {modeled:.3f} ms  vs  HIP events {r['ms']:.3f} ms (incl. finalize ~0.03)  ->  {100 * modeled / r['ms']:.1f} %  "
              f"[{groups} groups over {slots} slots]")
    print(f"#   the stamps themselves: stamped {two['ms']:.3f} ms vs product {plain_ms:.3f} ms = +{100 * (two['ms'] / plain_ms - 1):.1f} %")

    # ---- 3. measured cost model --------------------------------------------------------------------------------------------
    from scipy.optimize import nnls
    cols = ["mfma", "valu", "valu_pk", "valu_dpp", "lds_read", "lds_write", "vmem", "salu+waitcnt", "barrier"]
    A = np.array([[c["mfma"], c["valu"], c["valu_pk"], c["valu_dpp"], c["lds_read"], c["lds_write"], c["vmem"], c["salu"] + c["waitcnt"], 1.0]
                  for c in secs[:20]], dtype=np.float64)
    print("\n# Measured cost model: non-negative least squares of the 20 sections' SIMD time on their instruction counts (cycles per instruction).")
    print("#   1 WG/CU: a wave's own compute + wait per section;  2 WG/CU: (compute + wait) / 2 = the SIMD time one wave's section costs when two share the SIMD")
    for label, y in (("1 WG/CU", one["compute"] + one["wait"]), ("2 WG/CU", (two["compute"] + two["wait"]) / 2)):
        x, rn = nnls(A, y)
        fit = A @ x
        print(f"#   {label}: " + ", ".join(f"{n} {v:.1f}" for n, v in zip(cols, x)) + f";  rms residual {np.sqrt(np.mean((fit - y) ** 2)):.0f} cycles "
              f"({100 * np.sqrt(np.mean((fit - y) ** 2)) / y.mean():.1f} % of a mean section), worst section {int(np.argmax(np.abs(fit - y)))} "
              f"({(fit - y)[int(np.argmax(np.abs(fit - y)))]:+.0f})")
        # with the MFMA pinned at its hardware 32 cycles
        y2 = y - 32.0 * A[:, 0]
        x2, _ = nnls(A[:, 1:], np.maximum(y2, 0))
        fit2 = A[:, 1:] @ x2 + 32.0 * A[:, 0]
        print(f"#            with the MFMA pinned at 32: " + ", ".join(f"{n} {v:.1f}" for n, v in zip(cols[1:], x2))
              + f";  rms residual {np.sqrt(np.mean((fit2 - y) ** 2)):.0f}")

    # ---- 4. the idle share of the matrix pipe ----------------------------------------------------------------------------------
    mf = 32.0 * tot["mfma"]
    other_issue = 4.0 * (tot["valu"] + tot["valu_pk"] + tot["valu_dpp"] + tot["lds_read"] + tot["lds_write"] + tot["vmem"]) + tot["salu"] + tot["waitcnt"]
    t1 = one["compute"].sum() + one["wait"].sum() + one["between"]
    t2h = (two["compute"].sum() + two["wait"].sum() + two["between"]) / 2
    print("\n# Matrix pipe per wave and group: 32 x MFMA = %.0f cycles." % mf)
    print(f"#   1 WG/CU (one wave per SIMD): {t1:.0f} cycles per group -> pipe busy {100 * mf / t1:.1f} %.  Idle {t1 - mf:.0f} = barrier wait {one['wait'].sum():.0f} "
          f"+ vector / LDS / memory / scalar issue at 4 (1) cycles each {other_issue:.0f} + own stalls (LDS and memory latency, MFMA dependencies, s_waitcnt, back edge) "
          f"{t1 - mf - one['wait'].sum() - other_issue:.0f}")
    print(f"#   2 WG/CU (two waves per SIMD): {2 * t2h:.0f} cycles per group and wave, i.e. {t2h:.0f} of SIMD time each -> pipe busy {100 * mf / t2h:.1f} %.  "
          f"Idle {t2h - mf:.0f} = non-MFMA issue {other_issue:.0f} (if none of it overlapped an MFMA of the partner) + {t2h - mf - other_issue:.0f} "
          f"left over (arbitration, both waves stalled or parked at once)")
    print(f"#   while a wave waits at a barrier ({two['wait'].sum():.0f} of its {2 * t2h:.0f} cycles = {100 * two['wait'].sum() / (2 * t2h):.1f} %) its partner has the SIMD to itself")


if __name__ == "__main__":
    main()
