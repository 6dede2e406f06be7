"""configs[3] on ONE GPU, host-fed: writes a synthetic Map-free TEST split -- 130 scenes x 116 query frames
(every 5th of 580, lib/datasets/mapfree.py:383) with real 540x720 JPEG images and 16-bit millimetre depth PNGs, intrinsics.txt and
poses.txt in the dataset's own layout -- and runs submission.predict_fused over it: JPEG / PNG decode on the host, pinned batches,
H2D on a side stream, the fused pipeline, per-scene pose files, the zip.  Reports pairs/s, the fraction of the wall time the GPU
loop spent waiting for the loader, and what a resumed run costs.

python tools/bench_fused_split.py --root /tmp/mapfree_syn --scenes 130 --frames 116 --configs sg_pnp,loftr_emat --out profiles/r03_fused_split_1gpu.json
(the images of a scene are translated copies of one textured 3-band scene, so the matchers find real correspondences)"""
import argparse
import json
import os
import shutil
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_scene(args):
    root, s, frames = args
    from PIL import Image
    import mapfree_reloc_amd  # noqa: F401
    from mapfree_reloc_amd import images as IM
    d = os.path.join(root, "test", f"s{s:05d}")
    os.makedirs(os.path.join(d, "seq0"), exist_ok=True); os.makedirs(os.path.join(d, "seq1"), exist_ok=True)
    p = IM.synthetic_pair(100 + s)
    rgb = lambda im: np.repeat((np.clip(im, 0, 1) * 255 + 0.5).astype(np.uint8)[..., None], 3, 2)
    dpt = lambda z: (np.clip(z, 0, 65.0) * 1000 + 0.5).astype(np.uint16)
    Image.fromarray(rgb(p["img0"])).save(os.path.join(d, "seq0", "frame_00000.jpg"), quality=92)
    Image.fromarray(dpt(p["depth0"])).save(os.path.join(d, "seq0", "frame_00000.dptkitti.png"))
    lp = ["# name qw qx qy qz tx ty tz", "seq0/frame_00000.jpg 1 0 0 0 0 0 0"]
    lk = ["# name fx fy cx cy W H", "seq0/frame_00000.jpg 590.0 590.0 269.5 359.5 540 720"]
    for k in range(frames):
        fid = 5 * k
        # query frame k: the second view, nudged by a per-frame vertical roll of 8 px multiples so the files differ
        im = np.roll(p["img1"], 8 * (k % 5), axis=0); dz = np.roll(p["depth1"], 8 * (k % 5), axis=0)
        Image.fromarray(rgb(im)).save(os.path.join(d, "seq1", f"frame_{fid:05d}.jpg"), quality=92)
        Image.fromarray(dpt(dz)).save(os.path.join(d, "seq1", f"frame_{fid:05d}.dptkitti.png"))
    for fid in range(5 * frames):                     # poses / intrinsics list every frame; only every 5th is read (sample_factor 5)
        lp.append(f"seq1/frame_{fid:05d}.jpg 1 0 0 0 {-p['t_gt'][0]:.6f} 0 0")
        lk.append(f"seq1/frame_{fid:05d}.jpg 590.0 590.0 269.5 359.5 540 720")
    open(os.path.join(d, "poses.txt"), "w").write("\n".join(lp) + "\n")
    open(os.path.join(d, "intrinsics.txt"), "w").write("\n".join(lk) + "\n")
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default="/tmp/mapfree_syn")
    ap.add_argument("--scenes", type=int, default=130)
    ap.add_argument("--frames", type=int, default=116)
    ap.add_argument("--configs", default="sg_pnp,loftr_emat")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_fused_split_1gpu.json"))
    ap.add_argument("--writers", type=int, default=min(64, os.cpu_count() or 8))
    ap.add_argument("--graph", type=int, default=0, help="1: HIP.GRAPH_FUSED (matcher stage replayed from one HIP graph per batch shape)")
    ap.add_argument("--no-resume-legs", action="store_true")
    ap.add_argument("--decode", default="process", help="thread | process (HIP.LOADER_DECODE)")
    ap.add_argument("--workers", type=int, default=0, help="decode workers (HIP.LOADER_WORKERS; 0 = the granted CPUs)")
    ap.add_argument("--ref-cache", type=int, default=1, help="0: HIP.REF_FEATURE_CACHE off (SuperPoint on the reference view of every pair)")
    a = ap.parse_args()
    t0 = time.perf_counter()
    if not os.path.isdir(os.path.join(a.root, "test", f"s{a.scenes - 1:05d}")):
        shutil.rmtree(a.root, ignore_errors=True)
        with ProcessPoolExecutor(a.writers) as ex:
            list(ex.map(write_scene, [(a.root, s, a.frames) for s in range(a.scenes)]))
    t_write = time.perf_counter() - t0
    nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(a.root) for f in fs)

    import torch
    import mapfree_reloc_amd  # noqa: F401
    from mapfree_reloc_amd import submission
    from mapfree_reloc_amd.config import get_cfg_defaults
    res = {"tree": dict(scenes=a.scenes, pairs=a.scenes * a.frames, files_GB=round(nbytes / 1e9, 3), write_s=round(t_write, 1), writers=a.writers,
                        layout="test/sNNNNN/{seq0,seq1}/frame_*.jpg + .dptkitti.png, intrinsics.txt, poses.txt (540x720 JPEG q92, 16-bit mm PNG)"),
           "host_cores": os.cpu_count(), "gpu": torch.cuda.get_device_name(0)}
    for name in a.configs.split(","):
        cfg = get_cfg_defaults()
        cfg.DATASET.DATA_ROOT = a.root; cfg.DATASET.WIDTH = 540; cfg.DATASET.HEIGHT = 720; cfg.DATASET.ESTIMATED_DEPTH = "dptkitti"
        cfg.MODEL = "FeatureMatching"; cfg.ALLOW_SYNTHETIC_WEIGHTS = True
        cfg.HIP.GRAPH_FUSED = bool(a.graph)
        cfg.HIP.REF_FEATURE_CACHE = bool(a.ref_cache)
        cfg.HIP.LOADER_DECODE = a.decode
        cfg.HIP.LOADER_WORKERS = a.workers
        if name == "sg_pnp":
            cfg.FEATURE_MATCHING, cfg.POSE_SOLVER = "SuperGlue", "PNP"
            cfg.PNP.RANSAC_ITER, cfg.PNP.REPROJECTION_INLIER_THRESHOLD, cfg.PNP.CONFIDENCE = 1000, 3, 0.9999
            B = 32
        else:
            cfg.FEATURE_MATCHING, cfg.POSE_SOLVER = "LoFTR", "EssentialMatrixMetric"
            cfg.EMAT_RANSAC.PIX_THRESHOLD, cfg.EMAT_RANSAC.SCALE_THRESHOLD, cfg.EMAT_RANSAC.CONFIDENCE = 2.0, 0.1, 0.9999
            B = 16
        out_root = os.path.join(a.root, "out_" + name)
        shutil.rmtree(out_root, ignore_errors=True)
        # warm-up on two scenes (library initialisation, page cache stays as it is for the rest), then the whole split
        cfgw = cfg.clone(); cfgw.DATASET.SCENES = ["s00000", "s00001"]
        submission.predict_fused(cfgw, "test", os.path.join(a.root, "warm_" + name), batch_pairs=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = submission.predict_fused(cfg, "test", out_root, batch_pairs=B)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = dict(submission.LAST_RUN_STATS)
        base = dict(pairs=st["pairs"], seconds=round(dt, 2), pairs_per_s=round(st["pairs"] / dt, 1), batch_pairs=B, batches=st["batches"], graph=bool(a.graph),
                    loader_wait_s=round(st["loader_wait_s"], 2), loader_stall_fraction=round(st["loader_wait_s"] / st["seconds"], 4),
                    issue_s=round(st["issue_s"], 2), gpu_busy_s=round(st["gpu_busy_s"], 2), gpu_busy_fraction=round(st["gpu_busy_s"] / st["seconds"], 4),
                    decode_workers=st["decode_workers"], ref_feature_cache=bool(a.ref_cache), decode=a.decode, loader_stats={k: round(v, 3) for k, v in st.get("loader_stats", {}).items()},
                    phases_s=dict(predict_fused_call=round(dt, 2), inside_the_batch_loop=round(st["seconds"], 2), until_first_batch=round(st.get("first_batch_s", 0.0), 2),
                                  loop_incl_last_records=round(st.get("loop_s", 0.0), 2), loader_close=round(st.get("close_s", 0.0), 2)))
        if a.no_resume_legs:
            res[name] = base
            print(name, json.dumps(base), flush=True)
            continue
        # resume: every scene file present -> nothing is recomputed, the archive is rebuilt from the files
        t1 = time.perf_counter()
        z2 = submission.predict_fused(cfg, "test", out_root, batch_pairs=B)
        dt_resume = time.perf_counter() - t1
        st2 = dict(submission.LAST_RUN_STATS)
        # half-finished run: drop the files of the second half of the scenes, resume
        for s in range(a.scenes // 2, a.scenes):
            os.remove(os.path.join(out_root, "poses", f"pose_s{s:05d}.txt"))
        t2 = time.perf_counter()
        submission.predict_fused(cfg, "test", out_root, batch_pairs=B)
        dt_half = time.perf_counter() - t2
        st3 = dict(submission.LAST_RUN_STATS)
        import zipfile
        with zipfile.ZipFile(z) as zf:
            n_lines = sum(len(zf.read(nm).decode().strip().split("\n")) for nm in zf.namelist())
            n_files = len(zf.namelist())
        res[name] = dict(base, zip_scene_files=n_files, zip_pose_lines=n_lines,
                         resume_all_done=dict(seconds=round(dt_resume, 2), pairs_recomputed=st2["pairs"], same_zip=open(z, "rb").read() == open(z2, "rb").read()),
                         resume_half_done=dict(seconds=round(dt_half, 2), pairs_recomputed=st3["pairs"], scenes_recomputed=st3["scenes_computed"]))
        print(name, json.dumps(res[name]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
