"""Randomised parity soak of the polar filter (f-1; not part of the test suite): random polar clouds -- quantised intensities (ties
between bins, between lanes and between the 64-bin loads of a row), plateaus, long monotone runs, empty rows, rows at the +-pi
cut, random thresholds / transforms / row lengths / record layouts -- through randt_filter_scan_batch_dev and through the CPU
oracle; every output compared bit for bit (polar angle / peak angle to 1e-6: libm vs ocml atan2f).
Usage: filter_soak.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle as po  # noqa: E402
import randt_slam_amd as R  # noqa: E402
from randt_slam_amd import host  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
ctx = R.Context(0, torch.cuda.current_stream().cuda_stream)
bad = 0
n_points = n_rows_total = n_unorganised = 0
for case in range(n_cases):
    n_scans = int(rng.choice([1, 2, 5]))
    n_az = int(rng.choice([3, 12, 40, 130]))
    n_bins = int(rng.choice([17, 64, 100, 257, 770, 1500, 3100]))
    dr = float(rng.choice([0.02, 0.0438, 0.16]))
    layout = rng.choice(["xyzi", "xyiz", "pcl8", "five"])
    levels = int(rng.choice([3, 8, 50, 0]))                  # 0: continuous intensities
    kw = dict(min_range=float(rng.choice([0.0, 0.6, 2.5])), max_range=float(rng.choice([12.0, 40.0, 1e9])), min_intensity=float(rng.choice([0.0, 6.0, 30.0])),
              beam_thr=float(rng.choice([0.0, 0.04, 0.2, 100.0])))
    c, s_ = np.cos(rng.uniform(-3, 3)), np.sin(rng.uniform(-3, 3))
    T = np.array([[c, -s_, rng.uniform(-.2, .2), rng.uniform(-1, 1)], [s_, c, rng.uniform(-.2, .2), rng.uniform(-1, 1)], [0.01, -0.02, 1.0, rng.uniform(-1, 1)]], dtype=np.float32)
    kw["sensor_to_base"] = T
    scans = np.zeros((n_scans, n_az, n_bins, 4), np.float32)
    unorganised = -1                                         # the scan that got a foreign point, if any
    for s in range(n_scans):
        phase = rng.uniform(0, 2 * np.pi / n_az)
        az = -np.pi + phase + np.arange(n_az) * (2 * np.pi / n_az)
        if rng.random() < 0.3:
            az[int(rng.integers(0, n_az))] = np.pi - rng.uniform(0, 0.0015)      # a row the exact atan2f test has to walk
        r = (np.arange(n_bins) + 0.5) * dr
        I = rng.uniform(0, 20, (n_az, n_bins))
        for a in range(n_az):
            mode = rng.random()
            if mode < 0.15:
                I[a] = 0.0                                                       # nothing valid in this azimuth
            elif mode < 0.45:                                                    # a ramp: a long run on one or both sides
                cpk, half = int(rng.integers(0, n_bins)), int(rng.integers(1, 80))
                d = np.abs(np.arange(n_bins) - cpk)
                I[a] = np.where(d <= half, 60.0 - 0.3 * d, I[a])
            elif mode < 0.6:                                                     # a plateau (ties)
                cpk = int(rng.integers(0, n_bins))
                I[a, max(0, cpk - 3):cpk + 4] = 55.0
        if levels:
            I = np.round(I / 60.0 * levels) * (60.0 / levels)
        scans[s, ..., 0] = r[None] * np.cos(az)[:, None]
        scans[s, ..., 1] = r[None] * np.sin(az)[:, None]
        scans[s, ..., 2] = rng.uniform(-0.3, 0.3, (n_az, n_bins))
        scans[s, ..., 3] = I
    if rng.random() < 0.1 and n_az > 3:                                          # a foreign point inside a row: status 1 expected
        s, a, b = int(rng.integers(0, n_scans)), int(rng.integers(0, n_az - 1)), int(rng.integers(1, n_bins))
        scans[s, a, b, :2] = scans[s, (a + n_az // 2) % n_az, b, :2]
        unorganised = s
    elif rng.random() < 0.08 and n_az > 3 and n_bins > 2:                        # a zero-filled return inside a row: the reference's atan2(0, 0) = 0
        s, a, b = int(rng.integers(0, n_scans)), int(rng.integers(0, n_az - 1)), int(rng.integers(1, n_bins))   # starts a new azimuth there -> status 1
        if abs(float(np.arctan2(scans[s, a, 0, 1], scans[s, a, 0, 0]))) > 2e-4:   # (unless the row lies on the +x axis: angle 0 either way)
            scans[s, a, b, :2] = 0.0
            unorganised = s
    # record layouts: packed x y z I; packed x y I z (intensity in the third float); PCL's 8-float PointXYZI; 5 floats
    if layout == "xyzi":
        raw, ioff, oracle_in = scans, 3, scans
    elif layout == "xyiz":
        raw = scans[..., [0, 1, 3, 2]].copy()
        ioff, oracle_in = 2, raw
    elif layout == "pcl8":
        raw = np.zeros(scans.shape[:3] + (8,), np.float32)
        raw[..., :3], raw[..., 3], raw[..., 4] = scans[..., :3], 1.0, scans[..., 3]
        ioff, oracle_in = 4, scans
    else:
        raw = np.zeros(scans.shape[:3] + (5,), np.float32)
        raw[..., :3], raw[..., 4] = scans[..., :3], scans[..., 3]
        ioff, oracle_in = 4, scans
    oioff = 2 if layout == "xyiz" else 3
    fp, ofp = host.filter_params(**kw), po.filter_params(**kw)
    pitch = int(rng.choice([64, 4096, n_az * 200]))
    out = torch.zeros((n_scans, pitch, 4), dtype=torch.float32, device=dev)
    polar = torch.zeros((n_scans, pitch, 2), dtype=torch.float32, device=dev)
    peaks = torch.zeros((n_scans, n_az, 3), dtype=torch.float32, device=dev)
    counts = torch.zeros(n_scans, dtype=torch.int32, device=dev)
    pcounts = torch.zeros(n_scans, dtype=torch.int32, device=dev)
    status = torch.zeros(n_scans, dtype=torch.int32, device=dev)
    host.filter_scan_batch(ctx, torch.from_numpy(raw).to(dev), fp, out, counts, status, polar, peaks, pcounts, intensity_index=ioff)
    ctx.synchronize()
    out, polar, peaks, counts, pcounts, status = (t.cpu().numpy() for t in (out, polar, peaks, counts, pcounts, status))
    for s in range(n_scans):
        if s == unorganised:                                 # the reference would split that row in two: the HIP path says so instead
            n_unorganised += 1
            if status[s] not in (1, 2):                      # (2 = the output was too small as well: it takes precedence)
                bad += 1
                print("MISSED foreign point, case", case, "scan", s, int(status[s]))
            continue
        cnt, pts, pol, pk = po.filter_scan(oracle_in[s].reshape(-1, 4), ofp, ioff=oioff)
        n_points += cnt
        n_rows_total += n_az
        ok = True
        want_status = 2 if cnt > pitch else None
        emitted = min(cnt, pitch)
        ok &= counts[s] == emitted and pcounts[s] == len(pk)
        ok &= np.array_equal(out[s, :emitted].view(np.uint32), pts[:emitted].view(np.uint32))
        ok &= np.array_equal(polar[s, :emitted, 1], pol[:emitted, 1]) and np.allclose(polar[s, :emitted, 0], pol[:emitted, 0], atol=1e-6)
        g = peaks[s, :len(pk)]
        ok &= np.array_equal(g[:, 1:], pk[:, 1:]) and np.allclose(g[:, 0], pk[:, 0], atol=1e-6)
        if want_status is not None:
            ok &= status[s] == want_status
        else:
            ok &= status[s] == 0
        if not ok:
            bad += 1
            print("MISMATCH case", case, "scan", s, dict(n_az=n_az, n_bins=n_bins, layout=str(layout), levels=levels, pitch=pitch, cnt=cnt, got=int(counts[s]), status=int(status[s])),
                  {k: v for k, v in kw.items() if k != "sensor_to_base"})
print("filter soak: %d cases, %d azimuth rows, %d oracle points, %d mismatching scans, %d foreign-point clouds flagged" % (n_cases, n_rows_total, n_points, bad, n_unorganised))
sys.exit(1 if bad else 0)
