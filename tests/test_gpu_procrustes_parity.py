"""-m gpu parity: Procrustes path (csrc/procrustes.hip) vs the CPU oracle
(oracle/mfr_oracle_procrustes.c): bit-exact hypothesis counts, selected iteration, exit iteration,
inlier count and pose; known-answer recovery; plugin class."""
import numpy as np
import pytest
import torch

from mapfree_reloc_amd import solver_ops as ops, synth
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _case(n_list, seeds, outl, iters=4096, seed=0, depth_noise=0.002):
    batch = synth.make_batch(seeds, n_list, maxN=max(max(n_list), 4), outlier_frac=outl, noise_px=0.3, depth_noise=depth_noise,
                             zero_depth_frac=0.02)
    solver = ops.ProcrustesBatchSolver(0.05, 0.999, seed, iters)
    out = solver(_dev(batch["pts0"]), _dev(batch["pts1"]), _dev(batch["n_corr"]), _dev(batch["depth0"]), _dev(batch["depth1"]),
                 _dev(batch["K0"]), _dev(batch["K1"]), _dev(batch["pair_ids"]), diagnostics=True)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for b, n in enumerate(n_list):
        st, R, t, ninl = O.procrustes_solve(batch["pts0"][b, :n], batch["pts1"][b, :n], batch["depth0"][b], batch["depth1"][b],
                                            batch["K0"][b], batch["K1"][b], 0.05, 0.999, iters, seed, int(batch["pair_ids"][b]))
        assert out["status"][b] == st, (b, out["status"][b], st)
        assert out["n_inliers"][b] == ninl
        if st == 0:
            P, Q = O.procrustes_lift(batch["pts0"][b, :n], batch["pts1"][b, :n], batch["depth0"][b], batch["depth1"][b],
                                     batch["K0"][b], batch["K1"][b])
            ref = O.procrustes_ransac(P, Q, 0.05, 0.999, iters, seed, int(batch["pair_ids"][b]), want_counts=True)
            run = ref["iters_run"]
            np.testing.assert_array_equal(out["counts"][b, :run], ref["counts"][:run])
            assert out["best_iter"][b] == ref["best_iter"] and out["iters_run"][b] == run
            np.testing.assert_array_equal(out["R"][b], R)
            np.testing.assert_array_equal(out["t"][b], t.reshape(3))
        else:
            assert np.isnan(out["R"][b]).all()
    return batch, out


def test_procrustes_bit_exact_vs_oracle():
    _case([300, 1024, 3, 2, 0, 40, 2000], [1, 2, 3, 4, 5, 6, 7], outl=0.3)


def test_procrustes_outliers_seeds_budget():
    _case([800, 500], [11, 12], outl=0.6, seed=9)
    _case([600], [13], outl=0.5, iters=20)


def test_procrustes_known_answer():
    batch, out = _case([1500, 700], [21, 22], outl=0.3, depth_noise=0.0)
    for b in range(2):
        assert synth.rot_err_deg(out["R"][b], batch["R_gt"][b]) < 0.3
        assert np.linalg.norm(out["t"][b] - batch["t_gt"][b]) < 0.02
        assert out["n_inliers"][b] > 0.5 * batch["pairs"][b]["inlier_gt"].sum()


def test_procrustes_plugin_class():
    from mapfree_reloc_amd.config import get_cfg_defaults
    from mapfree_reloc_amd.matching.pose_solver import ProcrustesSolver
    cfg = get_cfg_defaults(); cfg.PROCRUSTES.MAX_CORR_DIST = 0.05
    p = synth.make_pair(31, 600, outlier_frac=0.3)
    data = {"depth0": torch.from_numpy(p["depth0"])[None], "depth1": torch.from_numpy(p["depth1"])[None],
            "K_color0": torch.from_numpy(p["K0"])[None], "K_color1": torch.from_numpy(p["K1"])[None], "pair_id": torch.tensor([31])}
    R, t, inl = ProcrustesSolver(cfg).estimate_pose(p["pts0"], p["pts1"], data)
    st, Rr, tr, ninl = O.procrustes_solve(p["pts0"], p["pts1"], p["depth0"], p["depth1"], p["K0"], p["K1"], seed=0, pair_id=31)
    assert inl == ninl and np.array_equal(R, Rr) and t.shape == (3, 1)
    cfg.PROCRUSTES.REFINE = True
    with pytest.raises(NotImplementedError):
        ProcrustesSolver(cfg)
