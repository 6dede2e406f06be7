"""Submission driver: same Pose line format, NaN filtering, per-scene grouping and zip layout as the
reference's submission.py:18-65 (`pose_{scene}.txt`, lines
`seq1/frame_XXXXX.jpg qw qx qy qz tx ty tz confidence`, README.md:182-212), on top of
build_model() -- plus a batched, pair-sharded fast path (predict_fused) that runs the fused GPU
pipeline and gathers the pose records over RCCL.

mat2quat is restated (transforms3d is not installed offline): w >= 0 convention.
"""
import argparse
from collections import defaultdict
from dataclasses import dataclass
from pathlib import Path
from zipfile import ZipFile

import numpy as np
import torch


def mat2quat(M):
    """rotation matrix -> (w,x,y,z), transforms3d.quaternions.mat2quat convention (largest-eigenvector
    method restated through the numerically stable pivot form; sign fixed so that w >= 0)"""
    from .parallel import rotmat_to_quat
    return rotmat_to_quat(torch.as_tensor(np.asarray(M, dtype=np.float64))).numpy()


@dataclass
class Pose:
    image_name: str
    q: np.ndarray
    t: np.ndarray
    inliers: float

    def __str__(self) -> str:
        formatter = {'float': lambda v: f'{v:.6f}'}
        max_line_width = 1000
        q_str = np.array2string(self.q, formatter=formatter, max_line_width=max_line_width)[1:-1]
        t_str = np.array2string(self.t, formatter=formatter, max_line_width=max_line_width)[1:-1]
        return f'{self.image_name} {q_str} {t_str} {self.inliers}'


def predict(loader, model):
    """per-pair loop of submission.py:33-58 (batch 1)"""
    results_dict = defaultdict(list)
    for data in loader:
        with torch.no_grad():
            R, t = model(data)
        R = R.detach().cpu().numpy()
        t = t.reshape(-1).detach().cpu().numpy()
        inliers = data['inliers']
        scene = data['scene_id'][0]
        query_img = data['pair_names'][1][0]
        if np.isnan(R).any() or np.isnan(t).any() or np.isinf(t).any():      # :48-49
            continue
        results_dict[scene].append(Pose(image_name=query_img, q=mat2quat(R).reshape(-1), t=t.reshape(-1), inliers=inliers))
    return results_dict


def records_to_results(records, id_to_name):
    """gathered [n,10] pose records (parallel.pose_records) -> results_dict; id_to_name maps
    pair_id -> (scene, query image name).  Failed pairs (NaN pose) are dropped like :48-49."""
    results = defaultdict(list)
    rec = np.asarray(records)
    for r in rec[np.argsort(rec[:, 0], kind="stable")]:
        if np.isnan(r[1:8]).any() or np.isinf(r[5:8]).any():
            continue
        scene, name = id_to_name[int(r[0])]
        results[scene].append(Pose(image_name=name, q=r[1:5].astype(np.float32), t=r[5:8].astype(np.float32), inliers=int(r[8])))
    return results


def save_submission(results_dict: dict, output_path: Path):
    with ZipFile(output_path, 'w') as zip:
        for scene, poses in results_dict.items():
            poses_str = '\n'.join((str(pose) for pose in poses))
            zip.writestr(f'pose_{scene}.txt', poses_str.encode('utf-8'))


def eval(args):
    """submission.py:68-91 on the datasets this repository can read (datasets.py)."""
    from .config import get_cfg_defaults
    from .builder import build_model
    from .datasets import make_loader
    cfg = get_cfg_defaults()
    if args.dataset_config:
        cfg.merge_from_file(args.dataset_config)
    cfg.merge_from_file(args.config)
    loader = make_loader(cfg, args.split)
    model = build_model(cfg, args.checkpoint)
    results_dict = predict(loader, model)
    args.output_root.mkdir(parents=True, exist_ok=True)
    save_submission(results_dict, args.output_root / 'submission.zip')


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('config', help='path to config file')
    parser.add_argument('--dataset_config', default='config/mapfree.yaml')
    parser.add_argument('--checkpoint', default='')
    parser.add_argument('--output_root', '-o', type=Path, default=Path('results/'))
    parser.add_argument('--split', choices=('val', 'test'), default='test')
    eval(parser.parse_args(argv))


if __name__ == '__main__':
    main()
