import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mapfree_reloc_amd
from mapfree_reloc_amd import synth
from mapfree_reloc_amd.matching import pose_solver as PS
from tools.bench_plugin import cfg_for
prs = [synth.make_pair(100 + i, 1024, outlier_frac=0.3) for i in range(20)]
datas = [{"depth0": torch.from_numpy(p["depth0"])[None], "depth1": torch.from_numpy(p["depth1"])[None],
          "K_color0": torch.from_numpy(p["K0"])[None], "K_color1": torch.from_numpy(p["K1"])[None], "pair_id": torch.tensor([i])} for i, p in enumerate(prs)]
for name, cls in (("PNP", PS.PnPSolver), ("EssentialMatrixMetric", PS.EssentialMatrixMetricSolver)):
    solver = cls(cfg_for("Precomputed", name))
    for p, d in zip(prs[:3], datas[:3]): solver.estimate_pose(p["pts0"], p["pts1"], d)
    torch.cuda.synchronize()
    # GPU-only time: events around the whole call sequence
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for p, d in zip(prs, datas): solver.estimate_pose(p["pts0"], p["pts1"], d)
    dt = time.perf_counter() - t0
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(name, "ms/pair", 1e3 * dt / 20); print(s.getvalue()[:2500])
