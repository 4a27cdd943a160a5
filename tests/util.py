"""Shared helpers for the parity tests: the same seeded problem is pushed through the CPU oracle
(oracle/pyoracle.py) and through the HIP library (randt_slam_amd, via the C ABI)."""
import functools

import numpy as np

import pyoracle as po
import randt_slam_amd as R
from randt_slam_amd import synth

IP = synth.indoor_params()


def oracle_map(cap=None):
    return po.Map(IP["size_x"], IP["size_y"], IP["resolution"], (0.0, 0.0), IP["max_neighbour_dist"],
                  IP["min_points_per_cell"], cap)


def oracle_scan_map(pts, cap=512):
    m = oracle_map(cap)
    m.build(pts, IP["n_clusters"], IP["max_range"], ioff=3 if pts.shape[1] == 4 else 4)
    return m


def oracle_submap(sm):
    """Rolling submap: merge the keyframe scans at their true relative poses (a9 + a18)."""
    sub = oracle_map()
    for t in range(len(sm["kf_scans"])):
        scan = oracle_scan_map(sm["kf_scans"][t])
        scan.transform(synth.pose3_to_pose4(sm["kf_rel"][t]))
        sub.merge(scan)
    return sub


@functools.lru_cache(maxsize=4)
def problem(n_submaps=2, scans_per_submap=8, n_keyframes=34, sigma=0.02):
    return synth.make_batch_problem(n_submaps=n_submaps, scans_per_submap=scans_per_submap, n_keyframes=n_keyframes,
                                    sigma=sigma)


def cells_equal(a, b):
    """Bit-exact comparison of two CELL_DTYPE arrays (ignoring the reserved word)."""
    if len(a) != len(b):
        return False
    for f in ("mean", "cov", "n", "max_intensity"):
        if not np.array_equal(np.asarray(a[f]).view(np.uint32), np.asarray(b[f]).view(np.uint32)):
            return False
    return True


def to_oracle_params(mp):
    """randt MatcherParams -> oracle MatcherParams (same fields; oracle adds linear_solver)."""
    op = po.default_params()
    for name, _ in R.MatcherParams._fields_:
        if name == "reserved":
            continue
        setattr(op, name, getattr(mp, name))
    return op


class GpuRig:
    """Device-side twin of a synth problem: submaps built by the HIP merge path, scans uploaded."""

    def __init__(self, prob, scan_cap=512):
        import torch

        self.torch = torch
        self.prob = prob
        self.dev = torch.device("cuda:0")
        self.ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
        self.mapp = R.indoor_map_params()
        self.clu = R.indoor_cluster_params()
        self.n_sub = len(prob["submaps"])
        self.B = len(prob["scans"])
        self.scan_cap = scan_cap
        self.submaps = R.Maps(self.ctx, self.n_sub, self.mapp, self.mapp.size_x * self.mapp.size_y, with_grid=True)
        self.scan_maps = R.Maps(self.ctx, self.B, self.mapp, scan_cap, with_grid=True)
        self.points = torch.from_numpy(prob["scans"]).to(self.dev)
        self.fixed_idx = torch.from_numpy(prob["submap_of"]).to(self.dev)

    def build_submaps(self):
        torch = self.torch
        for j, sm in enumerate(self.prob["submaps"]):
            kf = torch.from_numpy(np.stack(sm["kf_scans"])).to(self.dev)
            tmp = R.Maps(self.ctx, kf.shape[0], self.mapp, self.scan_cap, with_grid=False)
            R.ndt_build_batch(self.ctx, kf, self.clu, tmp)
            self.submaps.merge(j, tmp, 0, synth.pose3_to_pose4(sm["kf_rel"]))
            tmp.close()

    def build_scans(self):
        R.ndt_build_batch(self.ctx, self.points, self.clu, self.scan_maps)
        self.ctx.synchronize()
