"""Evaluation metrics (map-free-reloc_amd/evaluation.py) vs fixtures produced by the reference's own
benchmark code (oracle/gen_golden.py::gen_metrics), plus the known-answer cases of the reference's
benchmark/test_metrics.py (projection :164-174, identity reprojection :102-109, rotation error equals
the generating angle :60-83, small-angle accuracy of the sin variant :111-161)."""
import os

import numpy as np

from mapfree_reloc_amd import evaluation as E


def test_metrics_match_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_metrics.npz"))
    K, W, H = g["K"], int(g["W"]), int(g["H"])
    per_scene = {}
    for i in range(int(g["n_frames"])):
        p = f"f{i}_"
        m = E.frame_metrics(g[p + "q_est"], g[p + "t_est"], float(g[p + "conf"]), g[p + "q_gt"], g[p + "t_gt"], K, W, H)
        for k in ("trans_err", "rot_err", "reproj_err"):
            np.testing.assert_allclose(m[k], float(g[p + k]), rtol=1e-9, atol=1e-9)
        sc = per_scene.setdefault(int(g[p + "scene"]), {k: [] for k in m})
        for k, v in m.items():
            sc[k].append(v)
    agg = E.aggregate_results(per_scene, all_failures=5)
    assert len(agg) == int(g["n_agg"])
    for i, (k, v) in enumerate(agg.items()):
        assert k == str(g[f"agg{i}_name"])
        np.testing.assert_allclose(v, float(g[f"agg{i}_val"]), rtol=1e-9)
    qi, ti = E.convert_world2cam_to_cam2world(g["w2c_q"], g["w2c_t"])
    np.testing.assert_allclose(qi, g["c2w_q"], rtol=1e-12); np.testing.assert_allclose(ti, g["c2w_t"], rtol=1e-12)


def test_projection_known_answer():
    xyz = np.array(((10, 20, 30), (10, 30, 50), (-20, -15, 5), (-20, -50, 10)), dtype=np.float32)
    uv = np.array(((1 / 3, 2 / 3), (1 / 5, 3 / 5), (-4, -3), (-2, -5)), dtype=np.float32)
    assert np.allclose(uv, E.project(xyz, np.eye(3)))
    uv = np.array(((1 / 3, 2 / 3), (1 / 5, 3 / 5), (0, 0), (0, 0)), dtype=np.float32)
    assert np.allclose(uv, E.project(xyz, np.eye(3), img_size=(5, 5)))


def test_rotation_and_reprojection_properties():
    rng = np.random.default_rng(0)
    K = np.array([[590.0, 0, 269.5], [0, 590.0, 359.5], [0, 0, 1]])
    for _ in range(50):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        t = rng.normal(0, 10, 3)
        assert np.isclose(E.reproj_err(q, t, q, t, K, 540, 720), 0)
        ang = rng.uniform(-np.pi / 2, np.pi / 2)
        ax = rng.uniform(-1, 1, 3); ax /= np.linalg.norm(ax)
        dq = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax]
        assert np.isclose(E.rot_err(E.qmult(q, dq), q), abs(np.degrees(ang)), atol=1e-9)
        Rm = rng.normal(size=3)
        assert np.isclose(E.trans_err(t + Rm, t), np.linalg.norm(Rm))
    for scale in np.logspace(-1, -9, 9):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        ang = rng.uniform(-np.pi, np.pi) * scale
        ax = rng.uniform(-1, 1, 3); ax /= np.linalg.norm(ax)
        dq = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax]
        assert np.isclose(E.rot_err(E.qmult(q, dq), q), abs(np.degrees(ang)), rtol=0.1 * scale, atol=0.1 * scale)
