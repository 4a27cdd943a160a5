"""The reference's full SLAM loop as a call-pattern harness (row f-4): LocalFuser::processScan's graph bookkeeping
(src/local_fuser/local_fuser.cpp:164-223 keyframe node + odometry edge, :247-279 submap root node),
LocalFuser::detectLoopClosures (:318-410, Scan Context branch), NDTSlam::optimizePoseGraph (src/ndt_slam/ndt_slam.cpp:
351-361 -> GlobalFuser::optimizePoseGraph) and the pose part of LocalFuser::updateSubmaps (:65-88).

Everything numeric goes through the injected backend (odometry.HipBackend = the C ABI: NDT build, window registration,
Scan Context, pair registration, CS divergence, pose graph; tests inject an oracle backend with the same methods).  Not
built: OGM ray tracing / HierarchicalMap occupancy layers (SURVEY: out of scope), ROS timers (the caller decides when
to search and when to optimise)."""
import math

import numpy as np

from .odometry import Odometry, _se2_inv4, _se2_mul4

ODOM_SQRT_INFO = np.diag([10.0, 10.0, 50.0])          # local_fuser.cpp:203-205, :264-266


def _pose4(theta, x, y):
    return np.array([math.cos(theta), math.sin(theta), x, y])


def _angle(p4):
    return math.atan2(p4[1], p4[0])                    # so2().log()


class Slam(Odometry):
    keep_filtered_points = True

    def __init__(self, backend, matcher_params, window_params, loop_matcher_params, params=None, sc_params=None,
                 loop_closure_max_cs_divergence=3.6, loop_closure_weight=4.0e4, loop_sqrtI=None, pg_params=None):
        super().__init__(backend, matcher_params, window_params, params)
        self.loop_mp = loop_matcher_params
        self.max_cs = loop_closure_max_cs_divergence   # parameters_indoor.yaml:8
        self.loop_sqrt_info = loop_closure_weight * (np.eye(3) if loop_sqrtI is None else np.asarray(loop_sqrtI, dtype=np.float64))
        self.pg_params = dict(pg_params or {})
        backend.sc_open(dict(sc_params or {}))
        self.nodes = []                # global pose4 per node id (std::map<int, Pose>, keys 0..n-1)
        self.traversed = []            # Pose::traversed_dist
        self.edges = []                # (id_begin, id_end, trans pose4, sqrt_information 3x3)
        self.submap_idzs = []          # node id -> submap index
        self.root_nodes = {}           # submap index -> node id
        self.node_scans = {}           # scans_: node id -> scan handle (kept alive)
        self.submaps = {}              # submaps_: finished submap index -> submap handle
        self.pending_loop_search = []  # _next_maps_to_search_loop
        self.loop_log = []             # (query node, candidate node, cs divergence, accepted)
        self.n_optimizations = 0

    # ---- graph bookkeeping ------------------------------------------------------------------
    def _add_node(self, pose4, scan, points):
        nid = len(self.nodes)
        if nid > 0:                                                                       # :199-205, :258-267
            trans = _se2_mul4(_se2_inv4(self.nodes[nid - 1]), pose4)
            self.edges.append((nid - 1, nid, trans, ODOM_SQRT_INFO))
            dist = self.traversed[nid - 1] + float(np.hypot(trans[2], trans[3]))
        else:
            dist = 0.0
        self.nodes.append(np.array(pose4, dtype=np.float64))
        self.traversed.append(dist)
        self.submap_idzs.append(self.n_finished_submaps)
        self.node_scans[nid] = scan
        self._ref(scan)
        self.b.sc_append(points, pose4[2:], dist)                                         # :207, :281
        return nid

    def _on_first_scan(self, scan, points):
        nid = self._add_node(self.current_global_transform, scan, points)                 # :247-279
        self.root_nodes[self.n_finished_submaps] = nid

    def _on_keyframe(self, scan, points, smoothed_pose4):
        nid = self._add_node(_se2_mul4(self.current_global_transform, smoothed_pose4), scan, points)   # :192-222
        self.pending_loop_search.append(nid)

    def _on_submap_finished(self, submap):
        self.submaps[self.n_finished_submaps] = submap                                    # :43
        return True

    # ---- LocalFuser::detectLoopClosures, Scan Context branch (:318-350) ------------------------
    def detect_loop_closures(self):
        added = 0
        while self.pending_loop_search:
            q = self.pending_loop_search.pop(0)
            lid, yaw = self.b.sc_detect(q)
            if lid == -1 or self.submap_idzs[q] == self.submap_idzs[lid]:
                continue
            sub_i = self.submap_idzs[lid]
            if sub_i not in self.submaps:          # submaps_.at() would throw: the candidate's submap is still being built
                continue
            root = self.nodes[self.root_nodes[sub_i]]
            guess = _se2_mul4(_se2_mul4(_se2_inv4(root), self.nodes[lid]), _pose4(-yaw, 0.0, 0.0))      # :333
            est, _cost = self.b.register_pair(self.submaps[sub_i], self.node_scans[q], self.loop_mp, guess)   # :335
            cs = self.b.cs_divergence(self.submaps[sub_i], self.node_scans[q], est)                     # :338-339
            ok = bool(cs < self.max_cs)
            self.loop_log.append((q, lid, float(cs), ok))
            if ok:
                self.edges.append((self.root_nodes[sub_i], q, np.array(est, dtype=np.float64), self.loop_sqrt_info))   # :341-347
                added += 1
        return added

    # ---- NDTSlam::optimizePoseGraph (ndt_slam.cpp:351-361) --------------------------------------
    def optimize_pose_graph(self):
        if not self.nodes or not self.edges or self.submap_idzs[-1] <= 0:
            return None
        n_nodes_per_submap = math.ceil((self.submap_size_poses - (self.smoothing_steps - 1)) / self.insertion_step)
        max_update_index = int((len(self.nodes) - 1) / n_nodes_per_submap) * n_nodes_per_submap
        x = np.array([[p[2], p[3], _angle(p)] for p in self.nodes])
        ia = np.array([e[0] for e in self.edges], dtype=np.int32)
        ib = np.array([e[1] for e in self.edges], dtype=np.int32)
        meas = np.array([[e[2][2], e[2][3], _angle(e[2])] for e in self.edges])
        sqi = np.array([e[3] for e in self.edges])
        xo, res = self.b.pose_graph_optimize(x, ia, ib, meas, sqi, max_update_index, self.pg_params)
        for i in range(len(self.nodes)):
            self.nodes[i] = _pose4(xo[i, 2], xo[i, 0], xo[i, 1])                          # Sophus::SE2d(rot, pos), global_fuser.cpp:85
        # LocalFuser::updateSubmaps (:65-88), pose part: the current submap's origin follows its root node
        self.current_global_transform = self.nodes[self.root_nodes[self.n_finished_submaps]].copy()
        self.n_optimizations += 1
        return res

    def node_positions(self):
        return np.array([[p[2], p[3], _angle(p)] for p in self.nodes])
