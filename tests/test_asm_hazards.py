"""The fused kernel uses inline asm for packed adds; the compiler cannot pad MFMA -> VALU hazards inside
inline asm, so the device assembly is checked instead (tools/check_asm_hazards.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="no hipcc")
def test_no_inline_asm_reads_a_fresh_mfma_result():
    import check_asm_hazards
    n, bad = check_asm_hazards.check(os.path.join(ROOT, "hello_amd", "csrc", "readconv_fused.hip"))
    assert n > 0, "the inline-asm packed adds disappeared: update this check"
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="no hipcc")
def test_executed_mfma_accounting_matches_the_instruction_stream():
    """bench.py's executed-MFMA rate comes from readconv_pack.executed_macs_per_read: it must equal the v_mfma
    instructions the kernel really contains (one 16x16x4 MFMA = 1 024 MACs; per wave and group of 4 reads; the
    group loop is not unrolled)."""
    import check_asm_hazards
    from hello_amd import readconv_pack as rp
    path = os.path.join(ROOT, "hello_amd", "csrc", "readconv_fused.hip")
    cfg150 = "_ZN5hello15readconv_kernelINS_2rc3CfgILi4ELi4ELi150ELi0EEELb1ELi%dELb%dELb%dELb%dELb0EEEvNS_12ReadConvArgsE"
    for extra, wino in ((0, True), (2, True), (0, False)):
        per_wave_and_group = rp.executed_macs_per_read(wino, extra) * 4 / 4 / 1024       # 4 reads, 4 waves
        assert check_asm_hazards.mfma_count(path, cfg150 % (3 + extra, wino, 0, 0)) == per_wave_and_group, (extra, wino)
    # the arithmetic modes bf16x3 / bf16x3+32: 9 tiles x 6 chunks x 3 products per 64 -> 64 layer and wave (7 layers), 9 x 3 x 3
    # per 32 -> 32 layer (6 layers), on v_mfma_f32_16x16x32_bf16; the fp32 MFMAs of the layers they replace are gone
    fp32_total = int(rp.executed_macs_per_read(True, 0) * 4 / 4 / 1024)
    for with32, split, fp32_gone in ((0, 7 * 162, 7 * 240), (1, 7 * 162 + 6 * 81, 7 * 240 + 6 * 120)):
        asm = check_asm_hazards.assembly(path)
        sym = cfg150 % (3, 1, 1, with32)
        start = next(i for i, ln in enumerate(asm) if ln.strip().startswith(sym + ":"))
        end = next(i for i in range(start, len(asm)) if asm[i].strip().startswith("s_endpgm"))
        body = [ln.strip() for ln in asm[start:end]]
        assert sum(ln.startswith("v_mfma_f32_16x16x32_bf16") for ln in body) == split
        assert sum(ln.startswith("v_mfma_f32_16x16x4_f32") for ln in body) == fp32_total - fp32_gone
