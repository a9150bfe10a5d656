"""Instruction mix of the fused read-convolver kernel, per barrier-delimited section (one layer each), and the
issue-cycle estimate that goes with it.

On gfx950 fp32 MFMA and the other vector / LDS / memory instructions of a SIMD issue one after the other
(tools/mfma_valu_issue.hip): a v_mfma_f32_16x16x4_f32 takes 32 cycles, everything else ~4, scalar ~1.  The
estimate  32*MFMA + 4*(VALU + LDS + VMEM) + SALU  per wave and group of reads lands within 4 % of the measured
kernel time (2 waves per SIMD: time per group and workgroup slot = 2 x that, profiles/DESIGN_history_r01_r02.md section 7).

    python tools/instruction_mix.py [mangled-kernel-name-substring]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_asm_hazards as cah          # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = "_ZN5hello15readconv_kernelINS_2rc3CfgILi4ELi4ELi150ELi0EEELb1ELi3ELb1ELb0ELb0ELb0EEEvNS_12ReadConvArgsE"


def sections(symbol):
    lines = cah.assembly(os.path.join(ROOT, "hello_amd", "csrc", "readconv_fused.hip"))
    inside, cur, out = False, dict(mfma=0, valu=0, lds=0, vmem=0, salu=0), []
    for ln in lines:
        t = ln.strip()
        if not inside:
            inside = t.startswith(symbol + ":")
            continue
        op = t.split()[0] if t and t[0].isalpha() and not t.endswith(":") else ""
        if not op:
            continue
        kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_")
                else "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "salu")
        cur[kind] += 1
        if op == "s_barrier" or op == "s_endpgm":
            out.append(cur)
            cur = dict(mfma=0, valu=0, lds=0, vmem=0, salu=0)
        if op == "s_endpgm":
            break
    return out


def cycles(c):
    return 32 * c["mfma"] + 4 * (c["valu"] + c["lds"] + c["vmem"]) + c["salu"]


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    secs = sections(want)
    total = dict(mfma=0, valu=0, lds=0, vmem=0, salu=0)
    for i, c in enumerate(secs):
        for k in total:
            total[k] += c[k]
        est = cycles(c)
        print(f"section {i:2d}: mfma {c['mfma']:4d} valu {c['valu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} "
              f"salu {c['salu']:4d} | est {est:6d} cycles, MFMA share {100 * 32 * c['mfma'] / max(est, 1):3.0f} %")
    est = cycles(total)
    print(f"total     : mfma {total['mfma']} valu {total['valu']} lds {total['lds']} vmem {total['vmem']} salu {total['salu']}"
          f" | est {est} cycles per wave and group, MFMA share {100 * 32 * total['mfma'] / est:.1f} %")


if __name__ == "__main__":
    main()
