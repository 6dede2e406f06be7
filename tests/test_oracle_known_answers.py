"""Known-answer pins of the restated OpenCV routines in the CPU oracle (the reference's own tests
pin nothing on this path and OpenCV is not available offline: SURVEY.md 8c).  Synthetic two-view
geometry with known (R, t), inlier set and scale."""
import numpy as np

from mapfree_reloc_amd import synth
from oracle import oracle_lib as O


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def test_philox_known_answer_vectors():
    # Random123 kat_vectors (philox4x32-10)
    assert [hex(x) for x in O.philox([0, 0, 0, 0], 0, 0)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in O.philox([0xffffffff] * 4, 0xffffffff, 0xffffffff)] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], 0xa4093822, 0x299f31d0)] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_sampler_distinct_uniform():
    cnt = np.zeros(9)
    for it in range(6000):
        s = O.sample_distinct(3, 11, it, 9, 5)
        assert len(set(s.tolist())) == 5 and s.min() >= 0 and s.max() < 9
        cnt[s] += 1
    assert abs(cnt / cnt.sum() - 1 / 9).max() < 0.01
    assert sorted(O.sample_distinct(0, 0, 0, 5, 5).tolist()) == [0, 1, 2, 3, 4]


def test_det_log_and_iteration_cap():
    import math
    for x in [1e-4, 0.3, 0.999999, 2.5, 1e-300, 123456.789]:
        assert abs(O.det_log(x) - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))
    # OpenCV RANSACUpdateNumIters(p=.9999, ep, 4, 1000)
    for ep, want in [(0.5, 143), (0.2, 17), (0.0, 0), (1.0, 1000), (0.9, 1000)]:
        assert O.update_num_iters(0.9999, ep, 4, 1000) == want, ep


def test_polynomial_roots_wide_dynamic_range():
    rng = np.random.default_rng(1)
    for _ in range(400):
        nreal = int(rng.integers(0, 6)) * 2
        roots = list(rng.normal(size=nreal) * 10 ** rng.uniform(-1, 1.5, nreal))
        c = np.array([1.0])
        for r in roots:
            c = np.polymul(c, [1, -r])
        for _k in range((10 - nreal) // 2):
            a = rng.normal() * 3; b = abs(rng.normal()) * 3 + 0.01
            c = np.polymul(c, [1, -2 * a, a * a + b * b])
        c = c * 10 ** rng.uniform(-8, 8)
        got = O.poly_real_roots(c[::-1])
        np.testing.assert_allclose(got, np.sort(roots), rtol=1e-5, atol=1e-7)


def test_p3p_contains_ground_truth():
    rng = np.random.default_rng(2)
    for _ in range(300):
        R = synth.rand_rot(rng, 60); t = rng.uniform(-1, 1, 3)
        Xc = np.stack([rng.uniform(-2, 2, 3), rng.uniform(-2, 2, 3), rng.uniform(1, 8, 3)], 1)
        X = (R.T @ (Xc - t).T).T
        f = Xc / np.linalg.norm(Xc, axis=1, keepdims=True)
        Rs, ts = O.p3p(X, f)
        assert any(np.allclose(Ri, R, atol=1e-6) and np.allclose(ti, t, atol=1e-6) for Ri, ti in zip(Rs, ts))
        for Ri, ti in zip(Rs, ts):
            Y = (Ri @ X.T).T + ti
            np.testing.assert_allclose(Y / np.linalg.norm(Y, axis=1, keepdims=True), f, atol=1e-6)


def test_fivept_contains_ground_truth_and_satisfies_constraints():
    rng = np.random.default_rng(3)
    miss = 0
    for _ in range(300):
        R = synth.rand_rot(rng, 40); t = rng.normal(size=3); t /= np.linalg.norm(t)
        X = np.stack([rng.uniform(-2, 2, 5), rng.uniform(-2, 2, 5), rng.uniform(2, 8, 5)], 1)
        Y = (R @ X.T).T + t
        x0, x1 = X[:, :2] / X[:, 2:], Y[:, :2] / Y[:, 2:]
        Es = O.fivept(x0, x1)
        Eg = _skew(t) @ R; Eg /= np.linalg.norm(Eg)
        for E in Es:
            r = [np.r_[x1[i], 1] @ E @ np.r_[x0[i], 1] for i in range(5)]
            assert np.abs(r).max() < 1e-8
        miss += not any(min(np.abs(E - Eg).max(), np.abs(E + Eg).max()) < 1e-5 for E in Es)
    assert miss <= 9          # <= 3 %: Gauss-Jordan null space instead of SVD (documented)


def test_emat_decompose_twisted_pair():
    rng = np.random.default_rng(4)
    for _ in range(100):
        R = synth.rand_rot(rng, 90); t = rng.normal(size=3); t /= np.linalg.norm(t)
        E = _skew(t) @ R * rng.uniform(0.3, 3) * rng.choice([-1, 1])
        rc, Ra, Rb, tu = O.emat_decompose(E)
        assert rc == 0 and abs(abs(tu @ t) - 1) < 1e-9
        for Rk in (Ra, Rb):
            np.testing.assert_allclose(Rk @ Rk.T, np.eye(3), atol=1e-9)
            assert abs(np.linalg.det(Rk) - 1) < 1e-9
        assert min(np.abs(Ra - R).max(), np.abs(Rb - R).max()) < 1e-9


def test_pnp_and_emat_recover_known_pose():
    for n, outl in [(400, 0.3), (1500, 0.5)]:
        p = synth.make_pair(100 + n, n, outlier_frac=outl, noise_px=0.7)
        st, R, t, ninl = O.pnp_solve(p["pts0"], p["pts1"], p["depth0"], p["K0"], p["K1"], seed=0, pair_id=n)
        assert st == 0 and ninl > 0.8 * p["inlier_gt"].sum()
        assert synth.rot_err_deg(R, p["R_gt"]) < 0.2 and np.linalg.norm(t.ravel() - p["t_gt"]) < 0.02
        r = O.emat_solve(p["pts0"], p["pts1"], p["K0"], p["K1"], 2.0, seed=0, pair_id=n)
        tg = p["t_gt"] / np.linalg.norm(p["t_gt"])
        assert r["status"] == 0 and synth.rot_err_deg(r["R"], p["R_gt"]) < 0.5
        assert np.degrees(np.arccos(np.clip(r["t"] @ tg, -1, 1))) < 3.0
        assert (r["mask"].astype(bool) & ~p["inlier_gt"]).sum() <= 0.1 * r["n_inl"]       # few false inliers
        sc = O.scale_lift(p["pts0"], p["pts1"], r["mask"], p["depth0"], p["depth1"], p["K0"], p["K1"], r["R"], r["t"])
        cnt, bs, _ = O.scale_ransac(sc, 0.1)
        assert abs(bs - np.linalg.norm(p["t_gt"])) < 0.1 and cnt > 0.5 * r["n_inl"]


def test_solver_edge_cases():
    p = synth.make_pair(5, 50)
    assert O.pnp_solve(p["pts0"][:3], p["pts1"][:3], p["depth0"], p["K0"], p["K1"])[0] == O.ST_TOO_FEW
    assert O.pnp_solve(p["pts0"], p["pts1"], np.zeros_like(p["depth0"]), p["K0"], p["K1"])[0] == O.ST_BAD_DEPTH
    assert O.emat_solve(p["pts0"][:4], p["pts1"][:4], p["K0"], p["K1"])["status"] == O.ST_TOO_FEW
    st, R, t, n = O.pnp_solve(np.zeros((0, 2)), np.zeros((0, 2)), p["depth0"], p["K0"], p["K1"])
    assert st == O.ST_TOO_FEW and np.isnan(R).all() and n == 0


def test_procrustes_recovers_known_pose_and_kabsch_is_optimal():
    rng = np.random.default_rng(7)
    for n, outl in [(300, 0.3), (900, 0.6)]:
        R = synth.rand_rot(rng, 40); t = rng.normal(size=3)
        P = rng.uniform(-3, 3, (n, 3)) + np.array([0, 0, 6.0])
        Q = (R @ P.T).T + t + rng.normal(size=(n, 3)) * 0.01
        no = int(outl * n); Q[:no] += rng.normal(size=(no, 3)) * 2
        r = O.procrustes_ransac(P, Q, 0.05)
        assert r["status"] == 0 and r["n_inl"] >= 0.95 * (n - no)
        assert synth.rot_err_deg(r["R"], R) < 0.1 and np.linalg.norm(r["t"] - t) < 0.01
        # the final re-fit is the least-squares optimum on its inliers (compare with numpy SVD Kabsch)
        d = np.linalg.norm((r["R"] @ P.T).T + r["t"] - Q, axis=1)
        inl = d < 0.05
        Pc, Qc = P[inl] - P[inl].mean(0), Q[inl] - Q[inl].mean(0)
        U, _, Vt = np.linalg.svd(Qc.T @ Pc)
        Rk = U @ np.diag([1, 1, np.linalg.det(U @ Vt)]) @ Vt
        assert synth.rot_err_deg(r["R"], Rk) < 1e-3
    p = synth.make_pair(3, 800, outlier_frac=0.3)
    st, R, t, n = O.procrustes_solve(p["pts0"], p["pts1"], p["depth0"], p["depth1"], p["K0"], p["K1"])
    assert st == 0 and n > 300 and synth.rot_err_deg(R, p["R_gt"]) < 0.1 and np.linalg.norm(t.ravel() - p["t_gt"]) < 0.01
    assert O.procrustes_solve(p["pts0"][:2], p["pts1"][:2], p["depth0"], p["depth1"], p["K0"], p["K1"])[0] == O.ST_TOO_FEW
