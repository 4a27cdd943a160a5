"""Streaming-odometry harness: the call pattern of LocalFuser::processScan / initializeNewSubmap
(src/local_fuser/local_fuser.cpp:40-63,99-300) and NDTSlam::radarCb's submap roll-over
(src/ndt_slam/ndt_slam.cpp:211-223) on top of the C ABI -- BASELINE config 3.

Only the data path is reproduced (scan NDT -> predict -> fixed-lag registration -> keyframe queue
with insertion delay -> rolling submap merge -> new submap with overlap); pose-graph nodes,
ScanContext, ray tracing and ROS plumbing are out of scope.

The numerical work is done by a *backend*; the default one drives librandt_hip.so.  Tests inject
a backend built on the CPU oracle to check the whole sequence for parity -- this module itself never
imports the oracle.
"""
import numpy as np

from . import host
from ._capi import STATE_DTYPE
from .synth import indoor_params


def _se2_mul4(a, b):
    """Sophus SE2 product on [c, s, tx, ty] (double), complex re-normalised."""
    re, im = a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]
    n = np.hypot(re, im)
    return np.array([re / n, im / n, a[2] + a[0] * b[2] - a[1] * b[3], a[3] + a[1] * b[2] + a[0] * b[3]])


def _se2_inv4(a):
    c, s = a[0], -a[1]
    return np.array([c, s, -(c * a[2] - s * a[3]), -(s * a[2] + c * a[3])])


class HipBackend:
    """Device-resident maps in two pools: scan NDTs (ring of slots) and submaps."""

    def __init__(self, ctx, map_params, cluster_params, scan_capacity=512, scan_slots=24, submap_slots=4):
        import torch

        self.torch = torch
        self.ctx, self.mapp, self.clu = ctx, map_params, cluster_params
        self.scans = host.Maps(ctx, scan_slots, map_params, scan_capacity, with_grid=False)
        self.subs = host.Maps(ctx, submap_slots, map_params, map_params.size_x * map_params.size_y, with_grid=True)
        self.free_scans = list(range(scan_slots))
        self.free_subs = list(range(submap_slots))
        self._known_cells = {}      # submap slot -> a non-zero cell count seen earlier (see submap_cells)
        self.dev = torch.device("cuda", ctx.device)

    def _take(self, pool, name, knob):
        """Next free slot of a pool.  Slam keeps every keyframe scan and every finished submap referenced for loop
        closure (like scans_ / submaps_ of the reference's LocalFuser), so a SLAM run needs scan_slots >= number of
        keyframes + window and submap_slots >= number of submaps + 2."""
        if not pool:
            raise host.RandtError(3, "HipBackend", "%s slot pool exhausted: every slot is still referenced; create the backend "
                                  "with a larger %s" % (name, knob))
        return pool.pop(0)

    # ---- scans
    def build_scan(self, points):
        idx = self._take(self.free_scans, "scan", "scan_slots")
        pts = points if hasattr(points, "data_ptr") else self.torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(self.dev)
        host.ndt_build_batch(self.ctx, pts.reshape(1, pts.shape[-2], pts.shape[-1]), self.clu, self.scans, first_map=idx)
        return idx

    def build_scan_from_polar(self, raw, filter_params, pitch_out=6144):
        """BASELINE config 5 front end: RadarPreprocessor::filterScan on a raw polar scan
        (n_azimuths x n_bins x stride, device tensor) followed by the NDT build of the kept points."""
        torch = self.torch
        if getattr(self, "_f_out", None) is None or self._f_out.shape[1] != pitch_out:
            self._f_out = torch.zeros((1, pitch_out, 4), dtype=torch.float32, device=self.dev)
            self._f_cnt = torch.zeros(1, dtype=torch.int32, device=self.dev)
            self._f_status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        raw = raw if hasattr(raw, "data_ptr") else torch.from_numpy(np.ascontiguousarray(raw, dtype=np.float32)).to(self.dev)
        host.filter_scan_batch(self.ctx, raw.reshape(1, *raw.shape[-3:]), filter_params, self._f_out, self._f_cnt, self._f_status)
        idx = self._take(self.free_scans, "scan", "scan_slots")
        host.ndt_build_batch(self.ctx, self._f_out, self.clu, self.scans, first_map=idx, n_points=self._f_cnt)
        return idx

    def filtered_points(self):
        """The points filterScan kept for the last polar scan, as a host array (synchronises)."""
        n = int(self._f_cnt.cpu()[0])
        return self._f_out[0, :n].cpu().numpy()

    def release_scan(self, idx):
        self.free_scans.append(idx)

    # ---- submaps
    def new_submap(self):
        idx = self._take(self.free_subs, "submap", "submap_slots")
        self.subs.clear(idx, 1)
        self._known_cells.pop(idx, None)
        return idx

    def release_submap(self, idx):
        self.free_subs.append(idx)

    def submap_cells(self, idx):
        """Cell count of a submap (Map::isEmpty at local_fuser.cpp:123).  A submap only ever gains cells until it is cleared,
        so once a non-zero count has been read it is remembered: the per-scan "is the submap empty?" test then costs no
        device round trip (the read-back is a stream synchronisation, ~30 us per scan)."""
        n = self._known_cells.get(idx, 0)
        if n == 0:
            n = int(self.subs.counts(idx, 1)[0])
            if n > 0:
                self._known_cells[idx] = n
        return n

    def merge(self, sub_idx, scan_idx, pose4):
        self.subs.merge(sub_idx, self.scans, scan_idx, np.asarray(pose4, dtype=np.float64).reshape(1, 4))

    def copy_transformed(self, src_idx, pose4, reindex=False):
        """_last_submap_transformed = _current_submap; .transformMap(pose) (local_fuser.cpp:44-46).  reindex: also rebuild
        the index grid (the reference leaves it stale)."""
        dst = self._take(self.free_subs, "submap", "submap_slots")
        self._known_cells.pop(dst, None)
        self.subs.copy_from(self.subs, dst_first=dst, src_first=src_idx, count=1)
        self.subs.transform(dst, np.asarray(pose4, dtype=np.float64).reshape(1, 4))
        if reindex:
            self.subs.reindex(dst, 1)
        return dst

    # ---- matcher
    def predict(self, state, stamp, vector=False):
        return host.predict_state(state, stamp, host._capi.PARAM_VECTOR if vector else host._capi.PARAM_MANIFOLD)

    # ---- loop closure + back end (slam.py)
    def register_pair(self, sub_idx, scan_idx, mp, guess4):
        """Matcher::estimateLoopConstraint: submap (fixed) vs scan (moving); returns (pose4, cost)."""
        p, res = host.register_pair(self.ctx, self.subs, sub_idx, self.scans, scan_idx, mp, guess4)
        return p, float(res["cost"])

    def cs_divergence(self, sub_idx, scan_idx, pose4):
        return host.cs_divergence(self.ctx, self.subs, sub_idx, self.scans, scan_idx, pose4)[0]

    def sc_open(self, sp_kwargs):
        self._sc = host.ScDatabase(self.ctx, host.sc_params(**sp_kwargs))

    def sc_append(self, points, pos, dist):
        pts = points.cpu().numpy() if hasattr(points, "data_ptr") else points
        return self._sc.append(pts, pos, dist)

    def sc_detect(self, node_id):
        lid, yaw, _ = self._sc.detect(node_id)
        return lid, float(yaw)

    def pose_graph_optimize(self, x, ia, ib, meas, sqi, max_update_index, params_kwargs):
        return host.pose_graph_optimize(self.ctx, x, ia, ib, meas, sqi, max_update_index, host.pg_params(**params_kwargs))

    def register_window(self, fixed_idx, moving_idx, states, mp, wp, trans4, imu=None):
        st, t, rej, res = host.register_window(self.ctx, self.subs, fixed_idx, self.scans, moving_idx, states, mp, wp, trans4, imu)
        return st, t, rej, res


class Odometry:
    """LocalFuser's front-end loop.  process_scan(points, stamp) returns the scan's pose in the global
    frame as [c, s, tx, ty]."""

    def __init__(self, backend, matcher_params, window_params, params=None):
        p = dict(indoor_params())
        if params:
            p.update(params)
        self.b, self.mp, self.wp = backend, matcher_params, window_params
        self.insertion_step = p["insertion_step"]
        self.insertion_delay = p["smoothing_steps"] + 1              # ndt_slam.cpp:580
        self.smoothing_steps = p["smoothing_steps"]
        self.submap_size_poses = p["submap_size_poses"]
        self.submap_overlap = p["submap_overlap"]
        self.fix_submap_handover = bool(p.get("fix_submap_handover", False))   # False = the reference's behaviour
        self.vector = int(getattr(matcher_params, "parameterization", 0)) in (2, 3)   # VECTOR / ANALYTIC: (pos, rot) blocks (optimize_on_manifold: false or the analytic flag)
        self.current_submap = backend.new_submap()
        self.current_submap_is_empty = True                           # HierarchicalMap::is_empty (ndt_hierarchical_map.h:85-87): a flag, not a cell count
        self.last_submap_transformed = None
        self.trajectory = []                                          # list of STATE_DTYPE scalars
        self.map_window = []                                          # scan handles of the newest states
        self.next_maps_to_insert = []                                 # keyframe queue (scan handles)
        self.current_transform = np.array([1.0, 0.0, 0.0, 0.0])       # pose in the submap frame
        self.current_global_transform = np.array([1.0, 0.0, 0.0, 0.0])
        self.n_finished_submaps = 0
        self.last_state = None
        self.n_scans = 0
        self.n_registrations = 0
        self.n_rejected = 0
        self.last_result = None
        self.next_scans_to_insert = []                                # keyframe queue: the scans' points (Scan Context input)
        self._cur_points = None
        # Matcher::imu_constraints_ (ndt_matcher.cpp:18-20,22-59,360): one heading increment per predicted state, dropped at a
        # submap roll-over (local_fuser.cpp:51 resetMatcher); only read when window_params.use_imu is set
        self.imu_constraints = []
        self.initial_imu_bias = float(p.get("initial_imu_bias", 0.0))  # local_fuser.cpp:36,235
        self._yaw = 0.0

    # LocalFuser::getTransform (local_fuser.h:113-127)
    def get_transform(self):
        return _se2_mul4(self.current_global_transform, self.current_transform)

    def submap_complete(self):
        return len(self.trajectory) >= self.submap_size_poses

    # LocalFuser::initializeNewSubmap (local_fuser.cpp:40-63)
    def initialize_new_submap(self, initial_transform):
        b = self.b
        self.last_state = self.trajectory[-1].copy()
        if self.last_submap_transformed is not None:
            b.release_submap(self.last_submap_transformed)
        old_to_new = _se2_mul4(_se2_inv4(self.current_global_transform), initial_transform)   # :45, name and all
        if self.fix_submap_handover:
            # The reference's product is the NEW origin expressed in the OLD frame, i.e. the inverse of the map it is named
            # after, and transformMap leaves the index grid stale: the overlap map is misplaced unless the hand-over pose is
            # near identity (DESIGN "reference quirks").  Opt-in repair: apply the inverse and re-index.
            self.last_submap_transformed = b.copy_transformed(self.current_submap, _se2_inv4(old_to_new), reindex=True)
        else:
            self.last_submap_transformed = b.copy_transformed(self.current_submap, old_to_new)
        for h in self.next_maps_to_insert + self.map_window:
            self._unref(h)
        self.next_maps_to_insert, self.map_window, self.next_scans_to_insert = [], [], []
        self.imu_constraints = []                                      # :51 matcher_.resetMatcher()
        self.current_transform = np.array([1.0, 0.0, 0.0, 0.0])
        self.current_global_transform = np.array(initial_transform, dtype=np.float64)
        if not self._on_submap_finished(self.current_submap):          # submaps_.insert(...) (:43) keeps it alive
            b.release_submap(self.current_submap)
        self.current_submap = b.new_submap()
        self.current_submap_is_empty = True                            # :55 initialize()
        self.trajectory = []
        self.n_finished_submaps += 1

    # scan handles can sit in the window and in the keyframe queue at the same time
    def _ref(self, h):
        self._refs[h] = self._refs.get(h, 0) + 1

    def _unref(self, h):
        self._refs[h] -= 1
        if self._refs[h] == 0:
            del self._refs[h]
            self.b.release_scan(h)

    _refs = None
    keep_filtered_points = False

    # graph bookkeeping hooks of the SLAM layer (slam.py); the pure odometry loop does nothing here
    def _on_first_scan(self, scan, points):
        pass

    def _on_keyframe(self, scan, points, smoothed_pose4):
        pass

    def _on_submap_finished(self, submap):
        return False

    # LocalFuser::processScan (local_fuser.cpp:99-300), data path only
    def _process(self, scan, stamp):
        b = self.b
        # :108 `!_current_submap.isEmpty()`: HierarchicalMap's flag -- false from the first mergeMapCell on, even if that scan
        # produced no cell (ndt_hierarchical_map.cpp:68-72) -- not Map::isEmpty()'s cell count
        if not self.current_submap_is_empty:
            self.trajectory.append(b.predict(self.trajectory[-1], stamp, self.vector))    # :125 (predict / predictSE2 by optimize_on_manifold)
            self.imu_constraints.append(self._yaw)                                        # ndt_matcher.cpp:58 (inside predictTransform)
            self.map_window.append(scan)                                                  # :130
            self._ref(scan)
            fixed = [self.current_submap]
            if len(self.trajectory) < self.submap_overlap and self.n_finished_submaps > 0:  # :133-136
                fixed.append(self.last_submap_transformed)
            S = min(len(self.trajectory) - 1, self.smoothing_steps)                       # ndt_matcher.cpp:343
            states = np.array(self.trajectory[-S - 1:], dtype=STATE_DTYPE)
            imu = None
            if int(getattr(self.wp, "use_imu", 0)) and len(self.imu_constraints) > S:
                # imu_constraints_.end()[-i-1], i = S..1 -- sic, one step older than state i (ndt_matcher.cpp:360).  While the
                # vector is shorter than S + 1 the reference reads in front of its buffer (undefined); the window then runs
                # without IMU factors here and in the facade (DESIGN, spec decision 12)
                imu = [self.imu_constraints[-i - 1] for i in range(S, 0, -1)]
            states, trans, rej, res = (b.register_window(fixed, self.map_window[-S:], states, self.mp, self.wp, self.current_transform, imu)
                                       if imu is not None else
                                       b.register_window(fixed, self.map_window[-S:], states, self.mp, self.wp, self.current_transform))
            for j in range(S + 1):
                self.trajectory[len(self.trajectory) - S - 1 + j] = states[j]
            self.current_transform = trans
            self.n_registrations += 1
            self.n_rejected += int(bool(rej))
            self.last_result = res
            n = len(self.trajectory)
            if len(self.map_window) >= self.smoothing_steps:                              # :152-154
                self._unref(self.map_window.pop(0))
            if n % self.insertion_step == 0:                                              # :155-161 keyframe
                self.next_maps_to_insert.append(scan)
                self.next_scans_to_insert.append(self._cur_points)
                self._ref(scan)
            if n >= self.insertion_delay + self.insertion_step and (n - self.insertion_delay) % self.insertion_step == 0:  # :164
                smoothed = self.trajectory[-self.insertion_delay - 1]["pose"]             # :165-166
                kf = self.next_maps_to_insert.pop(0)
                b.merge(self.current_submap, kf, smoothed)                                # :177,190
                self._on_keyframe(kf, self.next_scans_to_insert.pop(0), smoothed)          # :192-222 node + edge
                self._unref(kf)
        else:
            # first scan of the submap (:225-295)
            st = np.zeros(1, dtype=STATE_DTYPE)[0]
            st["pose"] = self.current_transform
            st["pos"] = self.current_transform[2:]
            st["rot"] = np.arctan2(self.current_transform[1], self.current_transform[0])
            if self.n_finished_submaps == 0:
                st["imu_bias"] = self.initial_imu_bias                                    # :231-236
            else:
                st["lin_vel"], st["rot_vel"] = self.last_state["lin_vel"], self.last_state["rot_vel"]
                st["lin_acc"], st["imu_bias"] = self.last_state["lin_acc"], self.last_state["imu_bias"]
            st["stamp"] = stamp
            self.trajectory.append(st)
            self._on_first_scan(scan, self._cur_points)                                   # :247-279 root node of the submap
            b.merge(self.current_submap, scan, self.current_transform)                    # :281,293
            self.current_submap_is_empty = False

    def process_scan(self, points, stamp, polar_filter=None, imu_yaw_increment=0.0):
        """NDTSlam::radarCb (ndt_slam.cpp:211-223): process, roll the submap over when complete.
        polar_filter: FilterParams -> `points` is a raw polar scan and goes through filterScan first.
        imu_yaw_increment: the heading change since the last scan from the IMU (local_fuser.cpp:107-121), used when
        window_params.use_imu is set."""
        if self._refs is None:
            self._refs = {}
        self._yaw = float(imu_yaw_increment)
        self._cur_points = points
        if polar_filter is not None:
            scan = self.b.build_scan_from_polar(points, polar_filter)                     # :102 filterScan + clustering
            if self.keep_filtered_points:                                                 # the SLAM layer feeds them to Scan Context
                self._cur_points = self.b.filtered_points()
        else:
            scan = self.b.build_scan(points)                                              # :102-105
        self._ref(scan)
        self._process(scan, stamp)
        if self.submap_complete():
            self.initialize_new_submap(self.get_transform())
            self._process(scan, stamp)
        self._unref(scan)
        self.n_scans += 1
        return self.get_transform()


class ReplicaOdometry:
    """R independent copies of the front-end loop above advancing in LOCK-STEP on one GPU -- the sequential path's only scaling
    axis (scan t needs pose t-1, "replicas only"): R vehicles / R replays, each with its own scans, submaps and
    trajectory, sharing nothing.  All replicas follow the same schedule (keyframes, insertion delay and submap roll-over depend
    on scan counts only: local_fuser.cpp:152-164, ndt_slam.cpp:219-223), so one step is ONE launch per stage for all of them:
    randt_ndt_build_batch_dev (R scans), randt_register_window_batch (R windows, a workgroup each), randt_maps_merge_batch (R
    keyframe merges) -- where R Odometry objects need R contexts / streams and a host round trip each.
    Every replica computes exactly what an Odometry object computes on its scans (bit-identical: tests/test_gpu_window.py).

    Map layout: logical slot j of replica r is map j * R + r of the two shared batches, so that the maps one stage touches
    are contiguous over the replicas."""

    def __init__(self, ctx, n_replicas, map_params, cluster_params, matcher_params, window_params, params=None, scan_capacity=512,
                 scan_slots=24, submap_slots=4):
        import torch

        p = dict(indoor_params())
        if params:
            p.update(params)
        self.torch, self.ctx, self.R = torch, ctx, int(n_replicas)
        self.clu, self.mp, self.wp = cluster_params, matcher_params, window_params
        R = self.R
        self.scans = host.Maps(ctx, scan_slots * R, map_params, scan_capacity, with_grid=False)
        self.subs = host.Maps(ctx, submap_slots * R, map_params, map_params.size_x * map_params.size_y, with_grid=True)
        self.free_scans, self.free_subs = list(range(scan_slots)), list(range(submap_slots))
        self.insertion_step = p["insertion_step"]
        self.insertion_delay = p["smoothing_steps"] + 1
        self.smoothing_steps = p["smoothing_steps"]
        self.submap_size_poses, self.submap_overlap = p["submap_size_poses"], p["submap_overlap"]
        self.vector = int(getattr(matcher_params, "parameterization", 0)) in (2, 3)
        self.current_submap = self._new_submap()
        self.last_submap_transformed = None
        self.submap_known_nonempty = False
        self.trajectory = []                                   # list over time of (R,) STATE_DTYPE arrays
        self.map_window, self.next_maps_to_insert = [], []     # logical scan slots
        self.refs = {}
        ident = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (R, 1))
        self.current_transform, self.current_global_transform = ident.copy(), ident.copy()
        self.n_finished_submaps = 0
        self.last_state = None
        self.n_scans = self.n_registrations = self.n_rejected = 0
        self.last_results = None
        self._ar = np.arange(R, dtype=np.int32)

    # ---- slot bookkeeping (logical slots, shared by all replicas)
    def _new_submap(self):
        if not self.free_subs:
            raise host.RandtError(3, "ReplicaOdometry", "submap slot pool exhausted")
        j = self.free_subs.pop(0)
        self.subs.clear(j * self.R, self.R)
        return j

    def _ref(self, h):
        self.refs[h] = self.refs.get(h, 0) + 1

    def _unref(self, h):
        self.refs[h] -= 1
        if self.refs[h] == 0:
            del self.refs[h]
            self.free_scans.append(h)

    @staticmethod
    def _mul(a, b):       # Sophus SE2 product on (R, 4) arrays, complex re-normalised (the arithmetic of _se2_mul4, row by row)
        re, im = a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1], a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0]
        n = np.hypot(re, im)
        return np.stack([re / n, im / n, a[:, 2] + a[:, 0] * b[:, 2] - a[:, 1] * b[:, 3], a[:, 3] + a[:, 1] * b[:, 2] + a[:, 0] * b[:, 3]], 1)

    @staticmethod
    def _inv(a):
        c, s = a[:, 0], -a[:, 1]
        return np.stack([c, s, -(c * a[:, 2] - s * a[:, 3]), -(s * a[:, 2] + c * a[:, 3])], 1)

    def get_transform(self):
        return self._mul(self.current_global_transform, self.current_transform)

    def submap_complete(self):
        return len(self.trajectory) >= self.submap_size_poses

    def _submap_nonempty(self):
        # HierarchicalMap::isEmpty() is a flag that the first mergeMapCell clears whatever it merged (ndt_hierarchical_map.cpp:68-72,
        # local_fuser.cpp:108): a function of the scan COUNT, so the replicas cannot leave lock-step over it and nothing is read back
        return self.submap_known_nonempty

    def initialize_new_submap(self, initial_transform):
        R = self.R
        self.last_state = self.trajectory[-1].copy()
        if self.last_submap_transformed is not None:
            self.free_subs.append(self.last_submap_transformed)
        old_to_new = self._mul(self._inv(self.current_global_transform), initial_transform)        # local_fuser.cpp:45
        dst = self.free_subs.pop(0)
        self.subs.copy_from(self.subs, dst_first=dst * R, src_first=self.current_submap * R, count=R)   # :44
        self.subs.transform(dst * R, old_to_new)                                                      # :46 (index grid left stale)
        self.last_submap_transformed = dst
        for h in self.next_maps_to_insert + self.map_window:
            self._unref(h)
        self.next_maps_to_insert, self.map_window = [], []
        self.current_transform = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (R, 1))
        self.current_global_transform = np.array(initial_transform, dtype=np.float64)
        self.free_subs.append(self.current_submap)
        self.current_submap = self._new_submap()
        self.submap_known_nonempty = False
        self.trajectory = []
        self.n_finished_submaps += 1

    def _process(self, scan, stamp):
        R, ar = self.R, self._ar
        if self._submap_nonempty():
            last = self.trajectory[-1]
            par = host._capi.PARAM_VECTOR if self.vector else host._capi.PARAM_MANIFOLD
            self.trajectory.append(host.predict_states(last, stamp, par))      # Matcher::predictTransform, O(1) host math per replica
            self.map_window.append(scan)
            self._ref(scan)
            fixed = [self.current_submap]
            if len(self.trajectory) < self.submap_overlap and self.n_finished_submaps > 0:
                fixed.append(self.last_submap_transformed)
            S = min(len(self.trajectory) - 1, self.smoothing_steps)
            states = np.stack(self.trajectory[-S - 1:], axis=1)                 # (R, S + 1)
            fidx = np.stack([np.int32(j) * R + ar for j in fixed], 1)
            midx = np.stack([np.int32(j) * R + ar for j in self.map_window[-S:]], 1)
            states, trans, rej, res = host.register_window_batch(self.ctx, self.subs, fidx, self.scans, midx, states, self.mp, self.wp, self.current_transform)
            for j in range(S + 1):
                self.trajectory[len(self.trajectory) - S - 1 + j] = states[:, j].copy()
            self.current_transform = trans
            self.n_registrations += R
            self.n_rejected += int(rej.sum())
            self.last_results = res
            n = len(self.trajectory)
            if len(self.map_window) >= self.smoothing_steps:
                self._unref(self.map_window.pop(0))
            if n % self.insertion_step == 0:
                self.next_maps_to_insert.append(scan)
                self._ref(scan)
            if n >= self.insertion_delay + self.insertion_step and (n - self.insertion_delay) % self.insertion_step == 0:
                smoothed = self.trajectory[-self.insertion_delay - 1]["pose"]
                kf = self.next_maps_to_insert.pop(0)
                self.subs.merge_batch(self.current_submap * R, R, self.scans, kf * R, smoothed)
                self._unref(kf)
        else:
            st = np.zeros(R, dtype=STATE_DTYPE)
            st["pose"] = self.current_transform
            st["pos"] = self.current_transform[:, 2:]
            st["rot"] = np.arctan2(self.current_transform[:, 1], self.current_transform[:, 0])
            if self.n_finished_submaps > 0:
                for f in ("lin_vel", "rot_vel", "lin_acc", "imu_bias"):
                    st[f] = self.last_state[f]
            st["stamp"] = stamp
            self.trajectory.append(st)
            self.subs.merge_batch(self.current_submap * R, R, self.scans, scan * R, self.current_transform)
            self.submap_known_nonempty = True

    def process_scans(self, points, stamp):
        """points: (R, n_points, stride) float32 device tensor, scan r for replica r.  Returns the (R, 4) global poses."""
        if not self.free_scans:
            raise host.RandtError(3, "ReplicaOdometry", "scan slot pool exhausted")
        scan = self.free_scans.pop(0)
        host.ndt_build_batch(self.ctx, points, self.clu, self.scans, first_map=scan * self.R)
        self._ref(scan)
        self._process(scan, stamp)
        if self.submap_complete():
            self.initialize_new_submap(self.get_transform())
            self._process(scan, stamp)
        self._unref(scan)
        self.n_scans += 1
        return self.get_transform()
