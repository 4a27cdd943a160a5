"""CPU checks of the drop-in boundary: librandt_hip.so loads, exports every symbol include/randt.h
declares, the Python mirror's struct layouts equal the C ones, and compute entry points fail loudly
(no silent CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import randt_slam_amd as R
from randt_slam_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "randt.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(randt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/randt.h but not exported"
    assert set(names) == set(_capi.SYMBOLS), set(names) ^ set(_capi.SYMBOLS)
    assert lib.randt_version() == 100


def test_struct_layouts_match_c(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "randt.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(randt_cell), sizeof(randt_result),"
        " sizeof(randt_map_params), sizeof(randt_cluster_params), sizeof(randt_matcher_params),"
        " offsetof(randt_result, n_residuals), offsetof(randt_matcher_params, function_tolerance),"
        " offsetof(randt_cell, n), sizeof(randt_sc_params), offsetof(randt_sc_params, search_ratio),"
        " sizeof(randt_filter_params), sizeof(randt_bnb_params), sizeof(randt_pg_params), sizeof(randt_pg_result)); return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == 48 == _capi.CELL_DTYPE.itemsize
    assert out[1] == 64 == _capi.RESULT_DTYPE.itemsize
    assert out[2] == C.sizeof(_capi.MapParams)
    assert out[3] == C.sizeof(_capi.ClusterParams)
    assert out[4] == C.sizeof(_capi.MatcherParams)
    assert out[5] == _capi.RESULT_DTYPE.fields["n_residuals"][1]
    assert out[6] == _capi.MatcherParams.function_tolerance.offset
    assert out[7] == _capi.CELL_DTYPE.fields["n"][1]
    assert out[8] == C.sizeof(_capi.ScParams) and out[9] == _capi.ScParams.search_ratio.offset
    assert out[10] == C.sizeof(_capi.FilterParams) and out[11] == C.sizeof(_capi.BnbParams)
    assert out[12] == C.sizeof(_capi.PgParams) and out[13] == C.sizeof(_capi.PgResult)


def test_defaults_are_the_indoor_loop_closure_values():
    p = R.default_matcher_params()
    assert (p.loss_scale, p.loss_alpha, p.gnc_divisor, p.gnc_steps, p.n_neighbours) == (1.5, -2.0, 1.3, 2, 4)
    assert (p.max_iterations, p.function_tolerance, p.initial_radius) == (200, 1e-6, 1e4)
    assert p.parameterization == R.PARAM_AMBIENT4
    mp = R.indoor_map_params()
    assert (mp.size_x, mp.size_y, mp.resolution, mp.min_points_per_cell) == (100, 100, 0.5, 5)
    assert R.indoor_cluster_params().n_clusters == 2304


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(R.RandtError) as e:
        R.Context(0)
    assert e.value.status == _capi.ERR_NODEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "randt-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "randt_oracle" not in txt and "orc_" not in txt, f


def test_committed_counters_belong_to_the_current_kernels():
    """bench.py's roofline numerator comes from the newest profiles/r*_sq_summary.csv; its rows carry a fingerprint of the hot
    kernels' sources (tools/csrc_hash.py) and bench.py refuses counters taken from other code -- so a kernel change without
    tools/collect_profiles.sh + tools/pmc_summary.py fails HERE, on the CPU, before a stale number can be printed."""
    sys.path.insert(0, ROOT)
    import bench

    rows, path, stale = bench.load_counters()
    assert stale is None, stale
    assert set(rows) == set(bench.HOT_KERNELS), (path, sorted(rows))
