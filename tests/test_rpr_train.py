"""CPU: the regression training loop (regression/train.py; reference train.py:20-70, model.py:84-97,188-196) on the
Concat-aggregator configuration (the only aggregator without a device kernel): optimiser step, StepLR, gradient clipping,
Lightning-layout checkpoint + resume, and data-parallel training over gloo with world size 2 (replicas stay bit-identical,
gradients are the rank average).  The fused correlation-volume path trains in the -m gpu tests."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SETUP = r'''
import sys
sys.path.insert(0, %r)
import torch
from mapfree_reloc_amd.config import get_cfg_defaults
from mapfree_reloc_amd.regression.train import SyntheticPairs, Trainer
from oracle.gen_rpr_golden import CASES


def make(precision="fp32", clip=0.5):
    cfg = get_cfg_defaults()
    cfg.merge_from_list(CASES["concat_resnet"][0])
    cfg.merge_from_list(["TRAINING.LR", 1e-3, "TRAINING.LR_STEP_INTERVAL", 2, "TRAINING.LR_STEP_GAMMA", 0.5, "TRAINING.GRAD_CLIP", clip,
                         "TRAINING.PRECISION", precision, "TRAINING.EPOCHS", 1, "TRAINING.LOG_INTERVAL", 1, "TRAINING.VAL_INTERVAL", 1.0])
    return cfg
'''


def _ns():
    ns = {}
    exec(_SETUP % ROOT, ns)
    return ns


def test_train_steps_checkpoint_and_resume(tmp_path):
    ns = _ns()
    cfg = ns["make"]()
    src = ns["SyntheticPairs"](2, 64, 48, "cpu", seed=3)
    batches = [src.batch() for _ in range(5)]
    assert batches[0]["image0"].shape == (2, 3, 64, 48) and batches[0]["T_0to1"].shape == (2, 4, 4)
    R = batches[0]["T_0to1"][:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(2, 3, 3), atol=1e-5)
    tr = ns["Trainer"](cfg, "cpu", sample=batches[0]).build()
    p0 = [p.detach().clone() for p in tr.model.parameters()]
    losses = [tr.train_step(b)[2].item() for b in batches[:3]]
    assert all(l == l and abs(l) < 1e6 for l in losses)
    assert any(not torch.equal(a, b) for a, b in zip(p0, tr.model.parameters()))
    assert tr.global_step == 3 and abs(tr.opt.param_groups[0]["lr"] - 5e-4) < 1e-12          # StepLR stepped per optimiser step
    ck = tmp_path / "w" / "last.ckpt"
    tr.save(str(ck))
    saved = torch.load(ck, weights_only=True)
    assert set(saved) >= {"state_dict", "optimizer_states", "lr_schedulers", "epoch", "global_step"}
    assert all(k.split(".")[0] in ("encoder", "aggregator", "head", "s_r", "s_t") for k in saved["state_dict"])
    nxt = tr.train_step(batches[3])[2].item()
    tr2 = ns["Trainer"](cfg, "cpu", sample=batches[0])
    tr2.resume(str(ck))
    assert tr2.global_step == 3
    assert tr2.train_step(batches[3])[2].item() == nxt                                         # bit-identical continuation
    for a, b in zip(tr.model.parameters(), tr2.model.parameters()):
        assert torch.equal(a, b)
    summary = tr.validate(batches[3:])
    assert {"val_loss/loss", "val_metrics/R_err", "val_auc/pose_20", "val_t_scale/a1"} <= set(summary)


def test_bf16_autocast_step_and_fit(tmp_path):
    ns = _ns()
    cfg = ns["make"]("bf16", clip=0.0)
    src = ns["SyntheticPairs"](2, 256, 192, "cpu", seed=4)   # (mkldnn's bf16 weight gradient of a 3x3 conv over a 1x1 map is NaN: keep the head's maps > 1x1)
    tr = ns["Trainer"](cfg, "cpu", sample=src.batch())
    logs = []
    res = tr.fit(src, 2, [src.batch()], out_dir=str(tmp_path / "exp"), log=logs.append)
    assert all(p.dtype == torch.float32 for p in tr.model.parameters())                        # master weights stay fp32
    assert os.path.exists(tmp_path / "exp" / "last.ckpt") and os.path.exists(tmp_path / "exp" / "e0-last.ckpt")
    assert "val_loss/loss" in res and any("validation" in l for l in logs)


def test_flat_gradient_step_equals_the_standard_step():
    """TRAINING.GRAPH_STEP on a CPU device = the same step with gradients in one flat buffer and the clip on that buffer (the graph
    capture itself needs a GPU): same loss, same clipped gradients as the standard zero_grad / backward / clip_grad_norm_ step.
    (Parameters after Adam are not compared element by element: Adam turns a last-bit difference of a near-zero gradient into a
    +-lr step.)"""
    ns = _ns()
    src = ns["SyntheticPairs"](2, 128, 96, "cpu", seed=6)
    batches = [src.batch() for _ in range(2)]
    out = []
    for flat in (False, True):
        cfg = ns["make"]()
        cfg.TRAINING.GRAPH_STEP = flat
        tr = ns["Trainer"](cfg, "cpu", sample=batches[0]).build()
        assert tr.graph_step == flat
        loss = tr.train_step(batches[0])[2].item()
        out.append((loss, [p.grad.detach().clone() for p in tr.model.parameters()], tr))
    assert out[0][0] == out[1][0]
    gmax = max(float(g.abs().max()) for g in out[0][1])
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * gmax)
    assert all(p.grad.data_ptr() >= out[1][2]._flat.data_ptr() for p in out[1][2].model.parameters())       # still views of the flat buffer
    assert abs(out[1][2].train_step(batches[1])[2].item() - out[0][2].train_step(batches[1])[2].item()) < 1e-2


_WORKER = _SETUP + r'''
import torch.distributed as dist
cfg = make()
tr = Trainer(cfg, "cpu", sample=SyntheticPairs(2, 64, 48, "cpu", seed=0, rank=0).batch())
rank, world = tr.rank, tr.world
assert world == 2
tr.build()
src = SyntheticPairs(2, 64, 48, "cpu", seed=5, rank=rank)      # every rank trains on its own pairs
b = src.batch()
# gradient of one step == average of the two ranks' local gradients
tr.model.train()
R_loss, t_loss, loss = tr.step_mod(b)
loss.sum().backward()
g_ddp = torch.cat([p.grad.reshape(-1) for p in tr.model.parameters() if p.grad is not None]).clone()
tr.opt.zero_grad(set_to_none=True)
from mapfree_reloc_amd.regression.train import _Step
_Step(tr.model)(b)[2].sum().backward()                          # same replica, same batch, no all-reduce
g_loc = torch.cat([p.grad.reshape(-1) for p in tr.model.parameters() if p.grad is not None]).clone()
both = [torch.zeros_like(g_loc) for _ in range(2)]
dist.all_gather(both, g_loc)
assert torch.allclose(g_ddp, (both[0] + both[1]) / 2, rtol=1e-4, atol=1e-6), float((g_ddp - (both[0] + both[1]) / 2).abs().max())
assert not torch.allclose(both[0], both[1])
tr.opt.zero_grad(set_to_none=True)
for _ in range(3):
    tr.train_step(src.batch())
flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()])
parts = [torch.zeros_like(flat) for _ in range(2)]
dist.all_gather(parts, flat)
assert torch.equal(parts[0], parts[1]), "replicas diverged"
val = tr.validate([src.batch()])
assert "val_loss/loss" in val
tr.save(sys.argv[1])
# the flat-gradient mode (what TRAINING.GRAPH_STEP runs between replays): ONE all-reduce of one buffer instead of DDP's hooks
cfg2 = make(); cfg2.TRAINING.GRAPH_STEP = True
tr2 = Trainer(cfg2, "cpu", sample=SyntheticPairs(2, 64, 48, "cpu", seed=0, rank=0).batch()).build()
cfg3 = make()
tr3 = Trainer(cfg3, "cpu", sample=SyntheticPairs(2, 64, 48, "cpu", seed=0, rank=0).batch()).build()
src2 = SyntheticPairs(2, 64, 48, "cpu", seed=9, rank=rank)
bb = src2.batch()
tr2.train_step(bb); tr3.train_step(bb)
gm = max(float(p.grad.abs().max()) for p in tr3.model.parameters())
for a, c in zip(tr2.model.parameters(), tr3.model.parameters()):       # rank-averaged, clipped gradients of the two routes
    assert torch.allclose(a.grad, c.grad, rtol=1e-4, atol=1e-6 * gm), "flat-gradient step differs from DDP"
tr2.train_step(src2.batch())
f2 = torch.cat([p.detach().reshape(-1) for p in tr2.model.parameters()])
pp = [torch.zeros_like(f2) for _ in range(2)]
dist.all_gather(pp, f2)
assert torch.equal(pp[0], pp[1]), "flat-gradient replicas diverged"
tr2.validate([src2.batch()])
# what `bench.py --config rpr_train` runs by default on N ranks: both views in ONE encoder pass (per-view BatchNorm statistics) + the
# flat-gradient step.  Same rank-averaged gradients as the two-call DDP route, replicas stay identical.
cfg4 = make(); cfg4.TRAINING.GRAPH_STEP = True; cfg4.TRAINING.SIAMESE_BATCH = True
tr4 = Trainer(cfg4, "cpu", sample=SyntheticPairs(2, 64, 48, "cpu", seed=0, rank=0).batch()).build()
tr5 = Trainer(make(), "cpu", sample=SyntheticPairs(2, 64, 48, "cpu", seed=0, rank=0).batch()).build()
tr4.train_step(bb); tr5.train_step(bb)
gm = max(float(p.grad.abs().max()) for p in tr5.model.parameters())
for a, c in zip(tr4.model.parameters(), tr5.model.parameters()):
    assert torch.allclose(a.grad, c.grad, rtol=1e-3, atol=1e-5 * gm), "one-pass (siamese) flat-gradient step differs from the two-call DDP step"
tr4.train_step(src2.batch())
f4 = torch.cat([p.detach().reshape(-1) for p in tr4.model.parameters()] + [b.detach().float().reshape(-1) for b in tr4.model.buffers()])
p4 = [torch.zeros_like(f4) for _ in range(2)]
dist.all_gather(p4, f4)
assert torch.equal(p4[0][:f2.numel()], p4[1][:f2.numel()]), "siamese flat-gradient replicas diverged"
sys.stdout.write(f"rank {rank} ok {tr.global_step}\n"); sys.stdout.flush()
dist.destroy_process_group()
'''


def test_ddp_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", str(script), str(tmp_path / "ddp.ckpt")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok 3") == 2
    assert os.path.exists(tmp_path / "ddp.ckpt")


def _write_train_scene(root, name, n_frames, rng, n_pairs):
    """a Map-free TRAINING scene: two sequences, poses / intrinsics, overlaps.npz with (seqA, imA, seqB, imB) rows + scores"""
    import numpy as np
    from PIL import Image
    sc = root / name
    (sc / "seq0").mkdir(parents=True); (sc / "seq1").mkdir()
    lp, lk = ["# name qw qx qy qz tx ty tz"], ["# name fx fy cx cy W H"]
    for s in (0, 1):
        for i in range(n_frames):
            nme = f"seq{s}/frame_{i:05d}.jpg"
            Image.fromarray(rng.integers(0, 255, (96, 72, 3), dtype=np.uint8)).save(sc / nme, format="PNG")
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            lp.append(nme + " " + " ".join(f"{v:.8f}" for v in np.r_[q, rng.normal(size=3)]))
            lk.append(nme + " 100.0 110.0 35.5 47.5 72 96")
    (sc / "poses.txt").write_text("\n".join(lp) + "\n"); (sc / "intrinsics.txt").write_text("\n".join(lk) + "\n")
    idxs = np.stack([rng.integers(0, 2, n_pairs), rng.integers(0, n_frames, n_pairs), rng.integers(0, 2, n_pairs),
                     rng.integers(0, n_frames, n_pairs)], 1).astype(np.uint16)
    overlaps = rng.uniform(0.1, 1.0, n_pairs).astype(np.float32)
    np.savez(sc / "overlaps.npz", idxs=idxs, overlaps=overlaps)
    return idxs, overlaps


def test_training_split_reader_sampler_and_loader(tmp_path):
    """lib/datasets/mapfree.py:85-112 (overlaps.npz, overlap window), sampler.py (scene balance), datamodules.py:35-46: the
    reader on a synthetic training tree, the sampler's epoch dealt to two ranks, and one optimiser step from the loader"""
    import numpy as np
    from mapfree_reloc_amd import evaluation as E
    from mapfree_reloc_amd.datasets import MapFreeScene, SceneBalancedSampler, make_train_loaders
    ns = _ns()
    rng = np.random.default_rng(5)
    idxs, ov = _write_train_scene(tmp_path / "train", "s00001", 6, rng, 40)
    _write_train_scene(tmp_path / "train", "s00002", 5, rng, 25)
    # a val scene in the usual layout
    idv, _ = _write_train_scene(tmp_path / "val", "s00460", 6, rng, 1)
    os.remove(tmp_path / "val" / "s00460" / "overlaps.npz")
    sc = MapFreeScene(tmp_path / "train" / "s00001", (36, 48), 1, None, (0.4, 0.8))
    keep = idxs[(0.4 < ov) & (ov < 0.8)]
    assert len(sc) == len(keep) and sc.pairs == [tuple(int(v) for v in r) for r in keep]
    d = sc[3]
    sa, ia, sb, ib = keep[3]
    assert d["pair_names"] == (f"seq{sa}/frame_{ia:05d}.jpg", f"seq{sb}/frame_{ib:05d}.jpg") and d["pair_id"] == 3
    (q1, t1), (q2, t2) = sc.poses[d["pair_names"][0]], sc.poses[d["pair_names"][1]]
    q12 = E.qmult(q2, E.qinverse(q1))
    np.testing.assert_allclose(d["T_0to1"][:3, :3].numpy(), E.quat2mat(q12), atol=1e-6)
    np.testing.assert_allclose(d["T_0to1"][:3, 3].numpy(), t2 - E.rotate_vector(t1, q12), atol=1e-6)
    bw = MapFreeScene(tmp_path / "train" / "s00001", (36, 48), 1, None, (0.4, 0.8), black_white=True)[0]["image0"]
    assert torch.equal(bw[0], bw[1]) and torch.equal(bw[1], bw[2])
    # sampler: n per scene from every scene, identical epoch list on every rank, dealt round-robin
    s0, s1 = (SceneBalancedSampler([10, 4], 6, True, rank=r, world=2) for r in (0, 1))
    whole = SceneBalancedSampler([10, 4], 6, True).epoch_indices().tolist()
    a, b = list(s0), list(s1)
    assert a == whole[0::2] and b == whole[1::2] and sum(i < 10 for i in whole) == 6 and all(0 <= i < 14 for i in whole)
    nr = SceneBalancedSampler([10, 4], 6, False).epoch_indices().tolist()
    assert len(set(i for i in nr if i < 10)) == 6 and set(i for i in nr if i >= 10) == {10, 11, 12, 13}   # permutation, then padding
    # loader -> one optimiser step
    cfg = ns["make"]()
    cfg.merge_from_list(["DATASET.DATA_ROOT", str(tmp_path), "DATASET.HEIGHT", 128, "DATASET.WIDTH", 96, "DATASET.MIN_OVERLAP_SCORE", 0.2,
                         "DATASET.MAX_OVERLAP_SCORE", 0.9, "TRAINING.BATCH_SIZE", 2, "TRAINING.NUM_WORKERS", 0, "TRAINING.SAMPLER", "scene_balance",
                         "TRAINING.N_SAMPLES_SCENE", 4, "TRAINING.SAMPLE_WITH_REPLACEMENT", True])
    tl, vl = make_train_loaders(cfg, "cpu")
    assert len(tl) == 4                                        # 2 scenes x 4 samples / batch 2
    batch = next(iter(tl))
    assert batch["image0"].shape == (2, 3, 128, 96) and batch["T_0to1"].shape == (2, 4, 4) and batch["K_color0"].dtype == torch.float64
    tr = ns["Trainer"](cfg, "cpu", sample=batch).build()
    assert all(x == x for x in (v.item() for v in tr.train_step(batch)))
    assert sum(1 for _ in vl) == 1 and "val_loss/loss" in tr.validate(list(vl))       # 3 val pairs (every 5th of 6... ) -> drop_last


def test_multi_frame_query_reader(tmp_path):
    """lib/datasets/mapfree.py:273-368 + load_pairs' sample_offset branches: val/test windows of T consecutive frames every T+1, train
    rows filtered by window availability / map-frame-outside-window on the UNFILTERED valid-frame lists; image1 [T,3,H,W]"""
    import numpy as np
    from mapfree_reloc_amd.datasets import MapFreeSceneMultiFrame, list_scenes
    ns = _ns()
    rng = np.random.default_rng(8)
    T = 3
    idxs, ov = _write_train_scene(tmp_path / "train", "s00001", 8, rng, 60)
    sc = MapFreeSceneMultiFrame(tmp_path / "train" / "s00001", (36, 48), T, None, (0.3, 0.9))
    # brute-force restatement of the upstream comprehension
    idx64 = idxs.astype(np.int64)
    valid = {q: sorted(set(idx64[idx64[:, 0] == q, 1]) | set(idx64[idx64[:, 2] == q, 3])) for q in (0, 1)}
    want = []
    for (sa, ia, sb, ib), o in zip(idx64.tolist(), ov.tolist()):
        if not (0.3 < o < 0.9):
            continue
        k = valid[sb].index(ib) - T + 1
        if k < 0:
            continue
        w = tuple(int(v) for v in valid[sb][k:k + T])
        if sa != sb or ia < w[0] or ib < ia:
            want.append((sa, ia, sb, w))
    assert sc.pairs == want and len(want) > 5
    d = sc[2]
    assert d["image1"].shape == (T, 3, 48, 36) and d["image0"].shape == (3, 48, 36) and d["pair_id"] == 2 * (T + 1)
    assert d["pair_names"][1][-1] == f"seq{want[2][2]}/frame_{want[2][3][-1]:05d}.jpg" and len(d["pair_names"][1]) == T
    # val layout: windows end at positions T, 2T+1, ... of the sorted seq1 frames
    _write_train_scene(tmp_path / "val", "s00460", 9, rng, 1)
    os.remove(tmp_path / "val" / "s00460" / "overlaps.npz")
    cfg = ns["make"]()
    cfg.merge_from_list(["DATASET.DATA_ROOT", str(tmp_path), "DATASET.HEIGHT", 48, "DATASET.WIDTH", 36, "DATASET.QUERY_FRAME_COUNT", T])
    (val,) = list_scenes(cfg, "val")
    assert isinstance(val, MapFreeSceneMultiFrame) and [p[3] for p in val.pairs] == [(1, 2, 3), (5, 6, 7)]
    assert val.pair_name(1) == "seq1/frame_00007.jpg" and val[0]["image1"].shape == (T, 3, 48, 36)


def test_regression_model_through_the_submission_loop(tmp_path):
    """build_model(cfg) for MODEL 'Regression' + submission.predict / save_submission (submission.py:33-65 with data_to_model_device,
    lib/utils/data.py:4-17): every pair gets a finite pose line with confidence 0"""
    from mapfree_reloc_amd import submission
    from mapfree_reloc_amd.builder import build_model
    from mapfree_reloc_amd.datasets import make_loader
    ns = _ns()
    cfg = ns["make"]()
    cfg.merge_from_list(["DATASET.SYNTHETIC", [2, 3], "DATASET.HEIGHT", 256, "DATASET.WIDTH", 192])
    torch.manual_seed(0)
    model = build_model(cfg)
    assert not model.training
    res = submission.predict(make_loader(cfg, "val"), model)
    assert sorted(res) == ["s00000", "s00001"] and all(len(v) == 3 for v in res.values())
    z = tmp_path / "submission.zip"
    submission.save_submission(res, z)
    import zipfile
    with zipfile.ZipFile(z) as zf:
        lines = zf.read("pose_s00000.txt").decode().strip().splitlines()
    assert len(lines) == 3 and all(len(l.split()) == 9 and l.split()[-1] == "0" for l in lines)
    assert abs(sum(float(v) ** 2 for v in lines[0].split()[1:5]) - 1.0) < 1e-4          # unit quaternion


def test_siamese_batch_keeps_the_two_call_arithmetic():
    """TRAINING.SIAMESE_BATCH (both views of every pair in ONE encoder pass) with per-view BatchNorm statistics
    (encoder.ViewBatchNorm2d) == the reference's two encoder calls (lib/models/regression/model.py:64-66): same loss, same
    gradients, same running statistics and batch counters after the step."""
    ns = _ns()
    src = ns["SyntheticPairs"](3, 64, 48, "cpu", seed=5)
    b0 = src.batch()
    out = {}
    for siamese in (False, True):
        cfg = ns["make"]()
        cfg.merge_from_list(["TRAINING.SIAMESE_BATCH", siamese])
        torch.manual_seed(0)
        tr = ns["Trainer"](cfg, "cpu", sample=b0).build()
        tr.model.train()
        data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b0.items()}
        tr.model(data)
        loss = tr.model.loss_fn(data)[0] if isinstance(tr.model.loss_fn(data), (tuple, list)) else tr.model.loss_fn(data)
        loss.backward()
        out[siamese] = (loss.item(), {n: p.grad.clone() for n, p in tr.model.named_parameters() if p.grad is not None},
                        {n: b.clone() for n, b in tr.model.named_buffers()})
    (l0, g0, s0), (l1, g1, s1) = out[False], out[True]
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)), (l0, l1)
    assert g0.keys() == g1.keys()
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) <= 1e-4 * float(g0[n].abs().max()) + 1e-7, n
    for n in s0:
        if "num_batches_tracked" in n:
            assert int(s0[n]) == int(s1[n]) and int(s0[n]) % 2 == 0, n                  # two updates per forward in both paths
        else:
            assert torch.allclose(s0[n].float(), s1[n].float(), rtol=1e-5, atol=1e-7), n


def test_head_nan_flag_is_one_persistent_buffer_updated_in_place():
    """the sticky NaN flag must survive a captured training step -- one buffer whose address never changes, ORed in place by
    every forward and zeroed in place by the reader (rebinding the attribute would leave a graph replay writing into freed memory and
    switch the check off after the first read)"""
    import torch
    from mapfree_reloc_amd.regression.head import _Trunk
    t = _Trunk()
    assert "invalid" not in t.state_dict()                   # the reference's checkpoint keys are unchanged
    p = t.invalid.data_ptr()
    t._flag(torch.ones(3), torch.zeros(2))
    assert not bool(t.invalid) and t.invalid.data_ptr() == p
    t._flag(torch.tensor([1.0, float("nan")]))
    t._flag(torch.ones(3))                                   # sticky across later clean forwards
    assert bool(t.invalid) and t.invalid.data_ptr() == p
    t.clear_invalid()
    assert not bool(t.invalid) and t.invalid.data_ptr() == p
    t._flag(torch.tensor([float("inf")]))                    # ... and live again after the reader's reset
    assert bool(t.invalid)
