"""Offline matcher stage with the reference's CLI and output files
(etc/feature_matching_baselines/compute.py:13-44, 72-86): for every scene of the Map-free test+val
splits, match seq0/frame_00000.jpg against EVERY seq1 frame listed in poses.txt (not subsampled,
quirk Q4) and write `correspondences_{matcher}.npz` (key `correspondences`, NaN-padded
[Npairs, maxN, 4] float64) into the scene directory.

    python -m mapfree_reloc_amd.compute -ds Mapfree -m SG [--outdoor] [--data_root data/mapfree]
"""
import argparse
from pathlib import Path

from . import wire
from .matchers import MATCHERS


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', '-ds', type=str, default='Mapfree', choices=['Mapfree'])
    ap.add_argument('--matcher', '-m', type=str, default='SG', choices=MATCHERS.keys())
    ap.add_argument('--scenes', '-sc', type=str, nargs='*', default=None)
    ap.add_argument('--outdoor', action='store_true')
    ap.add_argument('--data_root', type=Path, default=Path('data/mapfree/'))
    args = ap.parse_args(argv)
    resize = 540, 720                                                       # compute.py:42
    matcher = MATCHERS[args.matcher](resize, args.outdoor)
    scenes = [f for split in ('test', 'val') if (args.data_root / split).is_dir()
              for f in sorted((args.data_root / split).iterdir()) if f.is_dir()]
    if args.scenes:
        scenes = [s for s in scenes if s.name in args.scenes]
    for scene_dir in scenes:
        qs = wire.parse_mapfree_query_frames(scene_dir / 'poses.txt')
        pts = [matcher.match((str(scene_dir / 'seq0' / 'frame_00000.jpg'), str(scene_dir / q))) for q in qs]
        wire.save_correspondences(scene_dir / f'correspondences_{args.matcher}.npz', pts)
        print(f'Finished {scene_dir.name}: {len(pts)} pairs')


if __name__ == '__main__':
    main()
