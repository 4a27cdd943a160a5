"""The hand-written 64-bit DPP instructions of the window solve (csrc/window.hip, pivot_group: `v_fmac_f64_dpp` /
`v_mov_b64_dpp` with `row_newbcast` inside inline assembly) are invisible to the compiler's hazard recogniser: on gfx9 a VALU
write of a VGPR needs two wait states before a DPP instruction reads it.  tools/check_dpp_hazards.py disassembles the built
object and checks every such instruction; this test keeps that check in the CPU suite (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_dpp_hazard_in_the_built_window_kernel(built):
    csrc = os.path.join(ROOT, "randt-slam_amd", "csrc")
    obj = os.path.join(csrc, "window.o")
    if not os.path.exists(obj):   # __graft_entry__.build() leaves the objects next to the sources; build this one if it has not run
        subprocess.check_call(["make", "-C", csrc, "window.o"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    n = int(r.stdout.split()[0])
    assert n >= 1000          # four instantiations of k_solve_window, hundreds of DPP updates each
