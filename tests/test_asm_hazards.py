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
