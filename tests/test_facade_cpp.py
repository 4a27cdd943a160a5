"""The C++ facade (include/randt_facade.hpp) compiles against the C ABI with plain g++ and behaves
like the reference classes: CPU box -> loud "no device" (exit 3); GPU box -> pose recovered (exit 0)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "randt-slam_amd")


def _build(tmp_path):
    exe = str(tmp_path / "facade_smoke")
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_smoke.cpp"),
        "-L", LIBDIR, "-lrandt_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe,
    ])
    return exe


def _has_gpu():
    import torch

    return torch.cuda.is_available()


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    if _has_gpu():
        pytest.skip("GPU present")
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    # default policy: no exception, a warning and a status; throwing policy: the same failure raises
    assert r.returncode == 3 and "no HIP device" in r.stdout and "WARNING: randt_ctx_create" in r.stdout


@pytest.mark.gpu
def test_facade_registers_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "WARNING: NO RESIDUALS ADDED!" in r.stdout
    assert "gnc_divisor" in r.stdout and "previous value kept" in r.stdout      # never-throw mode: warnings, not exceptions
