"""Static check of the fused kernel's device assembly: no inline-asm instruction may read a register that a
v_mfma wrote fewer wait states earlier than the hardware needs.

The MFMA -> VALU read-after-write hazard is software managed on gfx950: the compiler pads it with s_nop, but
its hazard recogniser does not look inside inline asm.  hello_amd/csrc/readconv_fused.hip uses inline asm for
v_pk_add_f32 (input transforms of the Winograd layers) and feeds it LDS-loaded operands only.  This script
counts wait states (an instruction = 1, `s_nop N` = N + 1) between every v_mfma write and every inline-asm read
of the same register and fails below REQUIRED (an 8-pass MFMA result needs 12 before a VALU read).  (A packed
inline-asm output transform behind an explicit `s_nop 11` was tried for the F(3,3) layers: bit-identical
results, 1.4 % slower than the compiler's scalar epilogue, which it schedules into the MFMA shadow.)

    python tools/check_asm_hazards.py [file.hip ...]      # exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"16x16": 12, "32x32": 20}     # wait states by MFMA shape (8 / 16 passes), with margin over the ISA's table


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"[va]\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"[va](\d+)$", tok)
    return {int(m.group(1))} if m else set()


_ASM = {}


def assembly(path):
    """Device assembly of ``path`` (compiled once per process), as a list of lines."""
    if path not in _ASM:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                            "--cuda-device-only", "-S", path, "-o", out], check=True, cwd=os.path.dirname(path),
                           stderr=subprocess.DEVNULL)
            _ASM[path] = open(out).read().split("\n")
    return _ASM[path]


def mfma_count(path, symbol):
    """v_mfma instructions in the body of the function whose mangled name is ``symbol``."""
    n, inside = 0, False
    for ln in assembly(path):
        t = ln.strip()
        if t.startswith(symbol + ":"):
            inside = True
        elif inside and t.startswith("s_endpgm"):
            return n
        elif inside and t.startswith("v_mfma"):
            n += 1
    raise KeyError(symbol)


def check(path):
    lines = assembly(path)
    last_writer, count, in_asm, bad, n_asm = {}, 0, False, [], 0      # register -> (instruction, index)
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not re.match(r"[a-z]", t):
            if t.endswith(":") and not t.startswith("."):
                last_writer = {}                      # a new function
            continue
        ops = t.split(None, 1)
        toks = ops[1].split(",") if len(ops) > 1 else []
        count += 1
        if ops[0] == "s_nop" and len(ops) > 1:
            count += int(ops[1].split()[0], 0)        # s_nop N = N + 1 wait states
        if in_asm and ops[0].startswith("v_"):
            n_asm += 1
            for x in toks[1:]:
                for r in regs(x.split()[0] if x.strip() else ""):
                    name, idx = last_writer.get(r, ("", -10 ** 9))
                    need = next((v for k, v in REQUIRED.items() if k in name), 20)
                    if name.startswith("v_mfma") and count - idx <= need:
                        bad.append((t, count - idx))
                        break
        if toks and not ops[0].startswith(("ds_write", "global_store", "buffer_store", "s_", "ds_store")):
            for r in regs(toks[0]):
                last_writer[r] = (ops[0], count)
    return n_asm, bad


def main():
    files = sys.argv[1:] or [os.path.join(ROOT, "hello_amd", "csrc", "readconv_fused.hip")]
    rc = 0
    for f in files:
        n, bad = check(os.path.abspath(f))
        print(f"{os.path.basename(f)}: {n} inline-asm vector instructions, {len(bad)} read a fresh MFMA result")
        for t, back in bad[:10]:
            print(f"   {t}    <- v_mfma {back} wait states earlier")
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main())
