"""Oracle-driven backend for randt_slam_amd.odometry.Odometry (test infrastructure): the same front-end
loop, every numeric step done by the CPU oracle, so whole drives can be compared pose by pose."""
import numpy as np

import pyoracle as po
from util import IP


class OracleBackend:
    def __init__(self, scan_capacity=512):
        self.scan_capacity = scan_capacity
        self.scans, self.subs = {}, {}
        self._next = 0

    def _map(self, cap=None):
        return po.Map(IP["size_x"], IP["size_y"], IP["resolution"], (0, 0), IP["max_neighbour_dist"], IP["min_points_per_cell"], cap)

    def build_scan(self, points):
        m = self._map(self.scan_capacity)
        pts = np.ascontiguousarray(points, dtype=np.float32)
        m.build(pts, IP["n_clusters"], IP["max_range"], ioff=3 if pts.shape[1] == 4 else 4)
        self._next += 1
        self.scans[self._next] = m
        return self._next

    def build_scan_from_polar(self, raw, filter_params, pitch_out=6144):
        import pyoracle as po_
        fp = po_.FilterParams()
        for name, _ in fp._fields_:
            v = getattr(filter_params, name)
            if name == "sensor_to_base":
                for i in range(12):
                    fp.sensor_to_base[i] = v[i]
            else:
                setattr(fp, name, v)
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        cnt, pts, _, _ = po_.filter_scan(raw.reshape(-1, raw.shape[-1]), fp, capacity=pitch_out)
        return self.build_scan(pts)

    def release_scan(self, h):
        del self.scans[h]

    def new_submap(self):
        self._next += 1
        self.subs[self._next] = self._map()
        return self._next

    def release_submap(self, h):
        del self.subs[h]

    def submap_cells(self, h):
        return self.subs[h].n_cells

    def merge(self, sub_h, scan_h, pose4):
        tmp = self.scans[scan_h].copy()
        tmp.transform(np.asarray(pose4, dtype=np.float64))
        self.subs[sub_h].merge(tmp)

    def copy_transformed(self, src_h, pose4):
        m = self.subs[src_h].copy()
        m.transform(np.asarray(pose4, dtype=np.float64))
        self._next += 1
        self.subs[self._next] = m
        return self._next

    def predict(self, state, stamp):
        return po.predict_state(np.asarray(state).astype(po.STATE_DTYPE), stamp)

    def register_window(self, fixed_h, moving_h, states, mp, wp, trans4):
        from test_gpu_window import to_oracle_wp
        from util import to_oracle_params

        rc, st, t, stats = po.register_window([self.subs[h] for h in fixed_h], [self.scans[h] for h in moving_h],
                                              np.asarray(states).astype(po.STATE_DTYPE), to_oracle_params(mp), to_oracle_wp(wp), trans4)
        assert rc >= 0
        return st, t, rc == 1, stats
