"""Submission driver.

Per-pair path (`predict`): the reference's loop (submission.py:33-58) over build_model(cfg)(data), batch 1.
Fused path (`predict_fused`): what replaces that serial loop on BASELINE configs[3] (full split, 8 GPUs) --
scenes sharded over the ranks in contiguous blocks balanced by pair count (parallel.shard_scenes), each rank
running pipeline.FusedPosePipeline on batches of pairs fed by a pinned-memory prefetching loader, one atomic
`pose_{scene}.txt` per finished scene (skip-if-present = per-scene resume), ONE all_gather of the 80-byte pose
records (RCCL on GPUs, gloo in the CPU tests) and the zip written by rank 0.

Output format (README.md:182-212, submission.py:18-30,60-65): zip of `pose_{scene}.txt`, lines
`seq1/frame_XXXXX.jpg qw qx qy qz tx ty tz confidence`, q and t with 6 decimals, pairs without a pose skipped.
mat2quat is restated (transforms3d is not installed offline): w >= 0 convention.
"""
import argparse
import json
import os
import time
from collections import defaultdict
from dataclasses import dataclass
from pathlib import Path
from zipfile import ZipFile, ZipInfo, ZIP_STORED

import numpy as np
import torch


def mat2quat(M):
    """rotation matrix -> (w,x,y,z), transforms3d.quaternions.mat2quat convention (largest-eigenvector
    method restated through the numerically stable pivot form; sign fixed so that w >= 0)"""
    from .parallel import rotmat_to_quat
    return rotmat_to_quat(torch.as_tensor(np.asarray(M, dtype=np.float64))).numpy()


def _fmt(v):
    return np.array2string(np.asarray(v), formatter={'float': lambda x: f'{x:.6f}'}, max_line_width=1000)[1:-1]


@dataclass
class Pose:
    """one line of pose_{scene}.txt (submission.py:18-30)"""
    image_name: str
    q: np.ndarray
    t: np.ndarray
    inliers: float

    def __str__(self) -> str:
        return f'{self.image_name} {_fmt(self.q)} {_fmt(self.t)} {self.inliers}'


def _has_pose(R, t):
    return not (np.isnan(R).any() or np.isnan(t).any() or np.isinf(t).any())       # submission.py:48-49


def data_to_model_device(data, model):
    """lib/utils/data.py:4-17: tensors follow the model's parameters; a model without parameters (the matching baselines: their
    solvers take host arrays) keeps the batch on the host"""
    try:
        device = next(model.parameters()).device
    except (StopIteration, AttributeError):
        device = torch.device('cpu')
    for k, v in data.items():
        if torch.is_tensor(v):
            data[k] = v.to(device)
    return data


def predict(loader, model):
    """scene -> [Pose] over a batch-1 loader; pairs without an estimate are left out"""
    results = defaultdict(list)
    for data in loader:
        data = data_to_model_device(data, model)                # submission.py:39
        with torch.no_grad():
            R, t = model(data)
        R, t = R.detach().cpu().numpy(), t.detach().cpu().numpy().reshape(-1)
        if not _has_pose(R, t):
            continue
        results[data['scene_id'][0]].append(Pose(data['pair_names'][1][0], mat2quat(R).reshape(-1), t, data['inliers']))
    return results


def records_to_results(records, id_to_name):
    """gathered [n,10] pose records (parallel.pose_records) -> results_dict; id_to_name maps the record id ->
    (scene, query image name).  Failed pairs (NaN pose) are dropped like submission.py:48-49."""
    results = defaultdict(list)
    rec = np.asarray(records)
    for r in rec[np.argsort(rec[:, 0], kind="stable")]:
        if not _has_pose(r[1:5], r[5:8]):
            continue
        scene, name = id_to_name[int(r[0])]
        results[scene].append(Pose(image_name=name, q=r[1:5].copy(), t=r[5:8].astype(np.float32), inliers=int(r[8])))
    return results


def scene_text(poses):
    return '\n'.join(str(p) for p in poses)


def save_submission(results_dict: dict, output_path: Path, deterministic=False):
    """zip of pose_{scene}.txt in dict order (submission.py:60-65).  deterministic=True pins the member timestamps so
    that equal results give byte-equal archives (used to compare world sizes)."""
    with ZipFile(output_path, 'w') as zf:
        for scene, poses in results_dict.items():
            text = (poses if isinstance(poses, str) else scene_text(poses)).encode('utf-8')
            name = f'pose_{scene}.txt'
            zf.writestr(ZipInfo(name, (1980, 1, 1, 0, 0, 0)) if deterministic else name, text, ZIP_STORED)


def _atomic_write(path: Path, text: str):
    tmp = path.with_name(path.name + f'.tmp{os.getpid()}')
    tmp.write_text(text, encoding='utf-8')
    os.replace(tmp, path)


LAST_RUN_STATS = {}          # predict_fused's timing of the last call in this process (pairs, seconds, loader_wait_s, ...)


def _run_signature(cfg, split):
    """hash of everything that determines the poses of a run: the merged configuration (dict order independent) and the split"""
    import hashlib

    def plain(v):
        if isinstance(v, dict):
            return {k: plain(v[k]) for k in sorted(v)}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        return v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)
    return hashlib.sha256(json.dumps({'cfg': plain(cfg), 'split': str(split)}, sort_keys=True).encode()).hexdigest()


def _rerun_out_of_range_scenes(pipeline, recs, names, scenes, todo, offsets, B, device, out_dir):
    """Second pass of predict_fused for the f16x2 range guard (pipeline.RangeGuard, include/mfr_hip.h mfr_f16x2_guard_bind): the hot loop never
    waits for the guard's flag -- a batch whose activations left the f16x2 range comes back with status ST_RANGE and NaN poses.  Here, after the loop,
    every scene that holds such a pair is computed AGAIN through the exact (bf16x3) twin of the pipeline, its rows replace the first pass's and its
    pose file is rewritten.  With in-range networks (every checkpoint seen so far) this function finds nothing and costs one comparison per row."""
    from . import options, parallel
    from .pipeline import ST_RANGE
    guard = getattr(pipeline, 'guard', None)
    if guard is None or not guard.active or not recs:
        return recs
    rows = np.concatenate(recs)
    hit = rows[:, 9] == ST_RANGE
    if not hit.any():
        return recs
    bad = {names[int(g)][0] for g in rows[hit, 0]}
    idx = [i for i in todo if scenes[i].scene_id in bad]
    from .datasets import PairBatchLoader, DevicePrefetcher
    keep = rows[np.array([names[int(g)][0] not in bad for g in rows[:, 0]], bool)]
    new_rows = []
    loader = PairBatchLoader([scenes[i] for i in idx], B, prefetch=1, pin=device.type == 'cuda', global_offsets=[int(offsets[i]) for i in idx], workers=2, decode='thread')
    try:
        with options.override(SPLIT='bf16x3'):
            twin = guard.twin()
            for batch in DevicePrefetcher(loader, device):
                out = twin(batch)
                new_rows.append(parallel.pose_records(batch['global_ids'].to(out['R'].device), out).cpu().numpy())
                guard.reruns += 1
    finally:
        loader.close()
    new_rows = np.concatenate(new_rows) if new_rows else np.zeros((0, parallel.REC_W))
    for sid, res in records_to_results(new_rows, names).items():
        _atomic_write(out_dir / f'pose_{sid}.txt', scene_text(res))
    return [keep, new_rows]


def predict_fused(cfg, split, output_root, pipeline=None, batch_pairs=None, resume=True, scenes=None, prefetch=2):
    """Scene-sharded, batched replacement of the reference's serial loop (submission.py:33-58) -> path of the zip
    (rank 0) or None (other ranks).  Works with or without an initialised torch.distributed process group
    (one process per GPU; backend nccl = RCCL on GPUs, gloo on CPU-only test boxes).

    Every rank: its contiguous scene block -> for each scene not yet on disk, batches of <= batch_pairs pairs through
    `pipeline` (default: FusedPosePipeline(cfg) on this rank's GPU; anything with `.device` and __call__(batch) -> dict(R, t,
    n_inliers, status) works) -> `pose_{scene}.txt` written atomically into output_root/poses (= the resume marker).
    Then ONE all_gather of the pose records of the pairs solved in this run; rank 0 assembles the zip in global scene
    order from those records (scenes finished in an earlier run: from their files)."""
    from . import options
    options.apply_cfg(cfg)
    import torch.distributed as dist
    from . import parallel
    from .datasets import list_scenes, PairBatchLoader, DevicePrefetcher
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    scenes = list(scenes) if scenes is not None else list_scenes(cfg, split)
    counts = [len(s) for s in scenes]
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    lo, hi = parallel.shard_scenes(counts, world)[rank] if scenes else (0, 0)
    out_dir = Path(output_root) / 'poses'
    out_dir.mkdir(parents=True, exist_ok=True)
    if pipeline is None:
        from .pipeline import FusedPosePipeline
        dev = torch.device('cuda', torch.cuda.current_device())
        pipeline = FusedPosePipeline(cfg, dev)
    device = torch.device(pipeline.device)
    B = int(batch_pairs or cfg.HIP.BATCH_PAIRS)

    # Resume is tied to WHAT produced the files: a manifest beside them holds a hash of the configuration (matcher, solver, thresholds,
    # weights paths, RANSAC seed, ...) and the split.  Pose files left by a different configuration -- or by a run that wrote no
    # manifest -- are stale: they are removed and recomputed, never mixed into the archive (the reference always recomputes).
    sig = _run_signature(cfg, split)
    mf = out_dir / 'manifest.json'
    if rank == 0:
        old = None
        if mf.exists():
            try:
                old = json.loads(mf.read_text()).get('signature')
            except (ValueError, OSError):
                old = None
        if not resume or old != sig:
            for f in out_dir.glob('pose_*.txt'):
                f.unlink()
        _atomic_write(mf, json.dumps({'signature': sig, 'split': split}))
    if world > 1:
        dist.barrier()
    todo = [i for i in range(lo, hi) if not (resume and (out_dir / f'pose_{scenes[i].scene_id}.txt').exists())]
    # decode threads: a pair costs ~45 ms of JPEG / PNG decode on one core (2 images + depth maps), the pipeline consumes 100-700 pairs/s
    # per GPU: the host's cores are split between the ranks of this node
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world)) if world > 1 else 1
    from .datasets import usable_cpus
    workers = max(2, min(32, usable_cpus() // max(1, local_world)))        # the CPUs the container grants, not the ones the box shows
    if 'LOADER_WORKERS' in cfg.HIP and int(cfg.HIP.LOADER_WORKERS) > 0:
        workers = int(cfg.HIP.LOADER_WORKERS)
    decode = str(cfg.HIP.LOADER_DECODE) if 'LOADER_DECODE' in cfg.HIP else 'process'
    loader = PairBatchLoader([scenes[i] for i in todo], B, prefetch=prefetch, pin=device.type == 'cuda',
                             global_offsets=[int(offsets[i]) for i in todo], workers=workers, decode=decode)
    recs, names, acc = [], {}, []
    stats = dict(pairs=0, batches=0, loader_wait_s=0.0, issue_s=0.0, gpu_busy_s=0.0, t0=time.perf_counter())
    it = iter(DevicePrefetcher(loader, device))
    # this thread issues ~500 small launches per batch while up to 32 decode threads want the interpreter lock for their bookkeeping
    # between two C calls: with CPython's default 5 ms switch interval every contended acquisition can cost this thread 5 ms and the
    # GPU runs dry (measured: 59 ms to issue a 39 ms step); 0.5 ms keeps the hand-over latency below a launch burst
    import sys
    from collections import deque
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(5e-4)
    # Finished scenes leave the loop WITHOUT draining the launch queue: their records go to a pinned host buffer by an asynchronous copy
    # and the files are written when the copy's event has completed (rounds 1-3 called .cpu() here: with a scene ending every 3-4
    # batches the GPU sat idle 17 % of the run while the host caught up after each drain).  Rows of scenes that are not complete yet
    # wait on the host side (host_rows), so nothing has to be filtered on the device.
    pending, host_rows, pinned = deque(), {}, []

    def land(block):
        while pending:
            ev, host, n, done = pending[0]
            if ev is not None:
                if block:
                    ev.synchronize()
                elif not ev.query():
                    return
            pending.popleft()
            srec = host[:n].numpy().copy()
            if ev is not None:
                pinned.append(host)
            for sid in {names[int(g)][0] for g in srec[:, 0]}:
                host_rows.setdefault(sid, []).append(srec[np.array([names[int(g)][0] == sid for g in srec[:, 0]], bool)])
            for sid in done:
                rows = np.concatenate(host_rows.pop(sid, [np.zeros((0, parallel.REC_W))]))
                _atomic_write(out_dir / f'pose_{sid}.txt', scene_text(records_to_results(rows, names).get(sid, [])))
                recs.append(rows)

    try:
        evs = []                                            # (start, end) event pairs around every step: GPU time of the steps, read at the end
        while True:
            tw = time.perf_counter()
            batch = next(it, None)                          # time blocked here = the GPU waiting for decode / H2D ("loader stall")
            stats['loader_wait_s'] += time.perf_counter() - tw
            if stats['batches'] == 0:
                stats['first_batch_s'] = time.perf_counter() - stats['t0']
            if batch is None:
                break
            ti = time.perf_counter()
            if device.type == 'cuda':
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            out = pipeline(batch)
            rec = parallel.pose_records(batch['global_ids'].to(out['R'].device), out)
            if device.type == 'cuda':
                e1.record(); evs.append((e0, e1))
            stats['issue_s'] += time.perf_counter() - ti    # host time to ISSUE the step (launches; any hidden synchronisation shows up here)
            acc.append(rec)
            sids = batch.get('scene_ids') or [batch['scene_id']] * len(batch['names'])
            for gid, sid, nm in zip(batch['global_ids'].tolist(), sids, batch['names']):
                names[gid] = (sid, nm)
            stats['pairs'] += len(batch['names']); stats['batches'] += 1
            done = batch.get('scenes_done')
            if done is None:
                done = [batch['scene_id']] if batch['last_of_scene'] else []
            if done:                                        # >= 1 scene complete -> one D2H copy of everything accumulated (batches may span scenes)
                dcat = torch.cat(acc)
                acc = []
                n = int(dcat.shape[0])
                if device.type == 'cuda':
                    if len(pending) >= 8:
                        land(True)                          # (never in practice: eight scene ends in flight)
                    host = next((h for h in pinned if h.shape[0] >= n), None)
                    if host is not None:
                        pinned.remove(host)
                    else:
                        host = torch.empty(max(n, 1024), parallel.REC_W, dtype=dcat.dtype, pin_memory=True)
                    host[:n].copy_(dcat, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    pending.append((ev, host, n, list(done)))
                else:
                    pending.append((None, dcat, n, list(done)))
            land(False)
        if acc:                                             # (cannot happen: the last pair of the rank's last scene ends a scene)
            pending.append((None, torch.cat(acc).cpu(), int(sum(a.shape[0] for a in acc)), []))
        land(True)
        for rows in host_rows.values():                     # (likewise: rows whose scene never reported its end)
            recs.extend(rows)
    except BaseException:
        # the pipeline or the loader raised: release the forked decode pool and the registered shared-memory ring, and land what finished scenes
        # already produced (their pose files are this run's resume markers) before the error propagates
        try:
            land(True)
        except Exception:
            pass
        loader.close()
        raise
    finally:
        sys.setswitchinterval(old_switch)
    stats['loop_s'] = time.perf_counter() - stats['t0']   # (includes landing the last scenes' records)
    if evs:
        torch.cuda.synchronize(device)
        stats['gpu_busy_s'] = sum(a.elapsed_time(b) for a, b in evs) * 1e-3
    tc = time.perf_counter()
    loader.close()                                          # decode processes / shared-memory slots of this run
    stats['close_s'] = time.perf_counter() - tc
    stats['seconds'] = time.perf_counter() - stats.pop('t0')
    LAST_RUN_STATS.clear(); LAST_RUN_STATS.update(stats, rank=rank, world=world, scenes_computed=len(todo), decode_workers=workers, decode=decode, batch_pairs=B, loader_stats=dict(getattr(loader, 'stats', {})))
    try:                                                    # one line per call and rank: the run's own record (pairs, seconds, stalls), next to the pose files
        line = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in LAST_RUN_STATS.items() if isinstance(v, (int, float, str, bool, dict))}
        line['unix_time'] = round(time.time(), 1)
        with open(out_dir / f'run_log_rank{rank}.jsonl', 'a', encoding='utf-8') as f:
            f.write(json.dumps(line, default=float) + '\n')
    except (OSError, TypeError, ValueError):
        pass
    recs = _rerun_out_of_range_scenes(pipeline, recs, names, scenes, todo, offsets, B, device, out_dir)
    mine = torch.from_numpy(np.concatenate(recs) if recs else np.zeros((0, parallel.REC_W))).to(device)
    allrec = parallel.gather_records(mine, world).cpu().numpy()
    if rank != 0:
        return None
    id_to_name = {}
    for si, sc in enumerate(scenes):                    # names of every pair (cheap: the naming rule, no IO)
        for k in range(len(sc)):
            id_to_name[int(offsets[si]) + k] = (sc.scene_id, sc.pair_name(k))
    fresh = records_to_results(allrec, id_to_name)
    solved_now = {id_to_name[int(g)][0] for g in allrec[:, 0]} if len(allrec) else set()
    ordered = {}
    for sc in scenes:
        if sc.scene_id in solved_now:
            if sc.scene_id in fresh:
                ordered[sc.scene_id] = scene_text(fresh[sc.scene_id])
        else:                                           # finished by an earlier run: take its file
            f = out_dir / f'pose_{sc.scene_id}.txt'
            text = f.read_text(encoding='utf-8') if f.exists() else ''
            if text:
                ordered[sc.scene_id] = text
    zpath = Path(output_root) / 'submission.zip'
    save_submission(ordered, zpath, deterministic=True)
    return zpath


def eval(args):
    """submission.py:68-91 on the datasets this repository can read (datasets.py)."""
    from .config import get_cfg_defaults
    cfg = get_cfg_defaults()
    if args.dataset_config:
        cfg.merge_from_file(args.dataset_config)
    cfg.merge_from_file(args.config)
    if args.synthetic:
        cfg.DATASET.SYNTHETIC = [int(v) for v in args.synthetic]
        cfg.DATASET.HEIGHT = cfg.DATASET.HEIGHT or 720
        cfg.DATASET.WIDTH = cfg.DATASET.WIDTH or 540
    args.output_root.mkdir(parents=True, exist_ok=True)
    if args.fused:
        import torch.distributed as dist
        launched = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
        if launched and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
            dist.init_process_group('nccl')
        predict_fused(cfg, args.split, args.output_root, batch_pairs=args.batch_pairs, resume=not args.no_resume)
        if launched:
            dist.destroy_process_group()
        return
    from .builder import build_model
    from .datasets import make_loader
    model = build_model(cfg, args.checkpoint)
    save_submission(predict(make_loader(cfg, args.split), model), args.output_root / 'submission.zip')


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('config', help='path to config file')
    parser.add_argument('--dataset_config', default=None, help="dataset yaml merged first (the reference hard-codes config/mapfree.yaml)")
    parser.add_argument('--checkpoint', default='')
    parser.add_argument('--output_root', '-o', type=Path, default=Path('results/'))
    parser.add_argument('--split', choices=('val', 'test'), default='test')
    parser.add_argument('--fused', action='store_true', help='batched, scene-sharded GPU path (one rank per GPU under torch.distributed.run)')
    parser.add_argument('--batch_pairs', type=int, default=None)
    parser.add_argument('--no_resume', action='store_true', help='recompute scenes whose pose file already exists')
    parser.add_argument('--synthetic', nargs=2, metavar=('N_SCENES', 'FRAMES'), default=None,
                        help='run on the synthetic stand-in dataset ON PURPOSE (no Map-free data offline)')
    eval(parser.parse_args(argv))


if __name__ == '__main__':
    main()
