#!/usr/bin/env python3
"""Static check of the hand-written DPP instructions in window.hip (band_pivot).

The fused `v_fmac_f64_dpp` updates are inline assembly, which the compiler's hazard recogniser cannot see into: on gfx9 a
VALU write of a VGPR must be followed by 2 wait states before a DPP instruction reads it.  This script disassembles the
object and verifies, for every 64-bit DPP instruction, that none of the instructions within the two preceding wait states writes
its DPP source registers.  Usage: tools/check_dpp_hazards.py [path/to/window.o]; exit code 1 on a violation."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def disassemble(obj):
    """obj: a single-translation-unit host object (window.o): its .hip_fatbin section holds one offload bundle."""
    fat, tmp = "/tmp/randt_dpp_check.fatbin", "/tmp/randt_dpp_check.co"
    subprocess.check_call([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([BUNDLER, "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + tmp],
                          stderr=subprocess.DEVNULL)
    return subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", tmp], text=True)


def check(text):
    ins = []
    for line in text.splitlines():
        line = line.split("//")[0].strip()
        if not line or line.endswith(":") or line.startswith(("/", ".", "Disassembly")):
            continue
        ins.append(line)
    n_dpp, bad = 0, []
    for i, line in enumerate(ins):
        if not line.startswith(("v_fmac_f64_dpp", "v_mov_b64_dpp")):
            continue
        n_dpp += 1
        ops = [t.strip() for t in line.split(None, 1)[1].split(",")]
        src = regs(ops[1].split()[0])   # the DPP operand is src0
        wait, j = 0, i - 1
        while wait < 2 and j >= 0:
            p = ins[j]
            if p.startswith("s_nop"):
                wait += int(p.split()[1], 0) + 1
            else:
                wait += 1
                if p.startswith("v_") and not p.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    dst = regs(p.split(None, 1)[1].split(",")[0].strip())
                    if dst & src:
                        bad.append((ins[j], line))
            j -= 1
    return n_dpp, bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "randt-slam_amd", "csrc", "window.o")
    n, bad = check(disassemble(lib))
    print("%d 64-bit DPP instructions (v_fmac_f64_dpp / v_mov_b64_dpp), %d hazard violations" % (n, len(bad)))
    for a, b in bad:
        print("  ", a, "->", b)
    sys.exit(1 if bad or n == 0 else 0)
