"""Row f-4 (evaluation half): trajectory text formats + drift evaluation, pinned to the result data the
reference repository ships for Oxford sequence 01 (tests/golden/oxford_eval_01.npz, made by
tests/golden/make_oxford_eval.py from oxford_results/randt_eval_16-13-09/slam)."""
import os

import numpy as np
import pytest

import randt_slam_amd as R
from randt_slam_amd import trajectory as tj

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oxford_eval_01.npz")


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)

    def rows(q):
        r = np.zeros((q.shape[0], 12))
        r[:, [0, 1, 3, 4, 5, 7]] = q / 1e6
        r[:, 10] = 1.0
        return r

    return dict(est=rows(z["est_micro"]), gt=rows(z["gt_micro"]), tum=[str(s) for s in z["tum_head"]], kitti=[str(s) for s in z["kitti_head"]],
                result=dict(zip([str(k) for k in z["result_keys"]], z["result_values"])),
                err_rows=z["errors_rows"], err_idx=z["errors_index"], err_n=int(z["errors_count"]))


def test_published_drift_numbers_are_reproduced(gold):
    ev = tj.evaluate(gold["gt"], gold["est"])
    ref = gold["result"]
    # result.txt prints 5-6 decimals
    assert abs(ev["trans_err_percent"] - ref["Trans.err.(%)"]) < 6e-6
    assert abs(ev["rot_err_deg_per_100m"] - ref["Rot.err.(deg/100m)"]) < 6e-6
    assert abs(ev["ate_m"] - ref["ATE(m)"]) < 6e-6
    assert abs(ev["rpe_m"] - ref["RPE(m)"]) < 6e-6 and abs(ev["rpe_dev_m"] - ref["RPE-dev(m)"]) < 6e-6
    assert abs(ev["rpe_deg"] - ref["RPE(deg)"]) < 6e-6 and abs(ev["rpe_dev_deg"] - ref["RPE-dev(deg)"]) < 6e-6
    assert abs(ev["bias_x_m"] - ref["bias-x(m)"]) < 6e-7 and abs(ev["bias_y_m"] - ref["bias-y(m)"]) < 6e-7
    assert abs(ev["rmse_m"] - ref["RMSE (m)"]) < 6e-6
    # per-segment rows of errors/01.txt (every 40th is in the fixture)
    seg = ev["segments"]
    assert seg.shape == (gold["err_n"], 5)
    assert np.allclose(seg[gold["err_idx"]], gold["err_rows"], rtol=0, atol=1e-12)


def test_kitti_rows_round_trip_the_reference_text(gold, tmp_path):
    head = gold["kitti"]
    rows = np.array([[float(v) for v in ln.split()] for ln in head])
    p4 = tj.kitti_to_pose4(rows)
    tj.write_kitti(tmp_path / "k.txt", p4)
    back = open(tmp_path / "k.txt").read().splitlines()
    for a, b in zip(back, head):
        va, vb = np.array(a.split(), float), np.array(b.split(), float)
        assert np.abs(va - vb).max() <= 1.5e-6          # the file's rotation block is only 6-decimal orthonormal
        assert [len(x.split(".")[1]) for x in a.split()] == [6] * 12
    assert np.allclose(tj.read_kitti(tmp_path / "k.txt"), p4, atol=2e-6)


def test_tum_rows_match_the_reference_format(gold):
    stamps, p4 = tj.read_tum(gold["tum"])
    for ln, t, p in zip(gold["tum"], stamps, p4):
        out = tj.format_tum_row(t, p)
        a, b = out.split(), ln.split()
        assert len(a) == len(b) == 8
        assert a[0].split(".")[0] == b[0].split(".")[0] and len(a[0].split(".")[1]) == 9   # %.9f stamp (float64 keeps ~1e-7 s)
        assert abs(float(a[0]) - float(b[0])) < 1e-6
        assert a[1:6] == b[1:6]                                                           # %.4f position, "0 0"
        assert abs(float(a[6]) - float(b[6])) <= 2e-3 * max(abs(float(b[6])), 1e-6) + 1e-9  # %.4g quaternion
        assert abs(float(a[7]) - float(b[7])) <= 1e-4


def test_formats_carry_device_poses(tmp_path):
    """poses in the C ABI layout [cos, sin, tx, ty] survive both formats."""
    th = np.linspace(-3, 3, 50)
    p4 = np.stack([np.cos(th), np.sin(th), np.linspace(-400, 400, 50), np.linspace(5, -900, 50)], axis=1)
    tj.write_kitti(tmp_path / "a.txt", p4)
    assert np.allclose(tj.read_kitti(tmp_path / "a.txt"), p4, atol=2e-6)
    tj.write_tum(tmp_path / "b.txt", np.arange(50) * 0.25, p4)
    st, q4 = tj.read_tum(str(tmp_path / "b.txt"))
    assert np.allclose(st, np.arange(50) * 0.25) and np.allclose(q4[:, 2:], p4[:, 2:], atol=1e-4)
    assert np.abs(np.arctan2(q4[:, 1], q4[:, 0]) - th).max() < 2e-3   # %.4g quaternion resolution
    ev = tj.evaluate(p4, p4)
    assert ev["ate_m"] < 1e-9 and ev["rpe_m"] < 1e-9 and (np.isnan(ev["trans_err_percent"]) or ev["trans_err_percent"] < 1e-9)
