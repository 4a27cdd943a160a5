"""Oracle-driven backend for randt_slam_amd.odometry.Odometry (test infrastructure): the same front-end
loop, every numeric step done by the CPU oracle, so whole drives can be compared pose by pose."""
import numpy as np

import pyoracle as po
from util import IP


class OracleBackend:
    def __init__(self, scan_capacity=512, params=None):
        self.scan_capacity = scan_capacity
        self.ip = dict(IP) if params is None else dict(params)   # map / clustering parameter set (default: indoor preset)
        self.scans, self.subs = {}, {}
        self._next = 0

    def _map(self, cap=None):
        ip = self.ip
        return po.Map(ip["size_x"], ip["size_y"], ip["resolution"], (0, 0), ip["max_neighbour_dist"], ip["min_points_per_cell"], cap)

    def build_scan(self, points):
        m = self._map(self.scan_capacity)
        pts = np.ascontiguousarray(points, dtype=np.float32)
        m.build(pts, self.ip["n_clusters"], self.ip["max_range"], ioff=3 if pts.shape[1] == 4 else 4)
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
        self._filtered = pts
        return self.build_scan(pts)

    def filtered_points(self):
        return self._filtered

    # ---- loop closure + back end (slam.py)
    def register_pair(self, sub_h, scan_h, mp, guess4):
        from util import to_oracle_params

        rc, pose, cost, _ = po.register_pair(self.subs[sub_h], self.scans[scan_h], to_oracle_params(mp), np.asarray(guess4, dtype=np.float64))
        return pose, cost

    def cs_divergence(self, sub_h, scan_h, pose4):
        m = self.scans[scan_h].copy()
        m.transform(np.asarray(pose4, dtype=np.float64))
        return po.cs_divergence(self.subs[sub_h], m)[0]

    def sc_open(self, sp_kwargs):
        d = dict(num_ring=20, num_sector=45, max_radius=15.0, num_exclude_recent=15, num_candidates=10, search_ratio=0.3,
                 dist_thresh=0.6, assumed_drift=0.05, odom_eps=1.2, odom_weight=0.2, intensity_factor=0.04)
        d.update(sp_kwargs)
        self._sp = po.ScParams(*[d[k] for k in ("num_ring", "num_sector", "max_radius", "num_exclude_recent", "num_candidates",
                                                  "search_ratio", "dist_thresh", "assumed_drift", "odom_eps", "odom_weight",
                                                  "intensity_factor")])
        self._sc_desc, self._sc_rk, self._sc_pos, self._sc_dist = [], [], [], []

    def sc_append(self, points, pos, dist):
        d, rk, _ = po.sc_make(np.ascontiguousarray(points, dtype=np.float32), self._sp)
        self._sc_desc.append(d)
        self._sc_rk.append(rk)
        self._sc_pos.append(np.array(pos, dtype=np.float64))
        self._sc_dist.append(float(dist))
        return len(self._sc_desc) - 1

    def sc_detect(self, node_id):
        lid, yaw, _ = po.sc_detect(self._sp, np.stack(self._sc_desc), np.stack(self._sc_rk), np.stack(self._sc_pos),
                                   np.array(self._sc_dist), node_id)
        return lid, float(yaw)

    def pose_graph_optimize(self, x, ia, ib, meas, sqi, max_update_index, params_kwargs):
        return po.pose_graph_optimize(x, ia, ib, meas, sqi, max_update_index, po.pg_params(**params_kwargs))

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

    def copy_transformed(self, src_h, pose4, reindex=False):
        m = self.subs[src_h].copy()
        m.transform(np.asarray(pose4, dtype=np.float64))
        if reindex:      # randt_maps_reindex: slots from the cells' current means, later cells win
            cells, grid = m.cells(), np.full(m.n_slots, -1, dtype=np.int32)
            for i, c in enumerate(cells):
                idx = po.lib().orc_map_coord_to_index(m._p, float(c["mean"][0]), float(c["mean"][1]))
                if idx < m.n_slots:
                    grid[idx] = i
            m.set(cells, grid)
        self._next += 1
        self.subs[self._next] = m
        return self._next

    def predict(self, state, stamp, vector=False):
        return po.predict_state(np.asarray(state).astype(po.STATE_DTYPE), stamp, vector=vector)

    def register_window(self, fixed_h, moving_h, states, mp, wp, trans4, imu=None):
        from test_gpu_window import to_oracle_wp
        from util import to_oracle_params

        rc, st, t, stats = po.register_window([self.subs[h] for h in fixed_h], [self.scans[h] for h in moving_h],
                                              np.asarray(states).astype(po.STATE_DTYPE), to_oracle_params(mp), to_oracle_wp(wp), trans4, imu)
        assert rc >= 0
        return st, t, rc == 1, stats
