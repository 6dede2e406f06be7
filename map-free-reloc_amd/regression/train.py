"""Training of the regression model: one process per GPU, data-parallel over RCCL, bf16 autocast.

Reference: train.py:20-70 (seed 0, Adam via `configure_optimizers`, gradient clipping, periodic validation, `last`
checkpoint, `--resume`), lib/models/regression/model.py:84-97 (training_step), :99-186 (validation), :188-196 (Adam
eps 1e-6 + StepLR stepped per optimiser step).  The reference runs `pl.Trainer(devices=1)`; here the ranks are explicit:

* every rank holds a replica; `torch.nn.parallel.DistributedDataParallel` all-reduces gradient BUCKETS
  (TRAINING.DDP_BUCKET_MB, default 64 MB: the model is ~25 MB of fp32 gradients, i.e. ONE ring all-reduce per step over
  xGMI, issued while the encoder's backward is still running) -- backend "nccl" is RCCL on ROCm, "gloo" on CPU;
* the loss is evaluated INSIDE the wrapped module's forward (`_Step`), because the heads publish their extra outputs
  (`q`, `R_bins`, `scale`, ...) by writing into the `data` dict and DDP hands the module a copy of that dict;
* TRAINING.PRECISION 'bf16': encoder / head convolutions and linears run under autocast(bfloat16) on the bf16 matrix
  cores, parameters / Adam state / BatchNorm statistics stay fp32, and the correlation-volume kernel, the Procrustes
  algebra and the losses compute in fp32 (they leave autocast themselves);
* validation batches are sharded over the ranks and the per-sample outputs gathered with one all_gather_object;
* checkpoints keep the Lightning layout the reference's `--resume` / `build_model(cfg, checkpoint)` read
  ({'state_dict', 'optimizer_states', 'lr_schedulers', 'epoch', 'global_step'}), written atomically by rank 0.

Data: `datasets.make_train_loaders` reads the Map-free train split (overlaps.npz pair lists, overlap window,
scene-balanced sampling dealt round-robin to the ranks, DataLoader workers, pinned batches uploaded one step ahead).  No
dataset is reachable offline, so the bench and the tests use `SyntheticPairs`: seeded image pairs with a known relative
pose and the same `data` keys (image0, image1, T_0to1)."""
import argparse
import contextlib
import math
import os
import time

import torch
import torch.distributed as dist
import torch.nn as nn

from .model import RegressionModel, RegressionMultiFrameModel


# ----------------------------------------------------------------------------------------------------------------
# distributed plumbing
# ----------------------------------------------------------------------------------------------------------------
def init_distributed(device=None):
    """-> (rank, world, device).  Under torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK in the env) joins the
    process group: RCCL when a HIP device is used, gloo on CPU."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device is None:
        device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    if device.type == "cuda":
        if local >= torch.cuda.device_count():
            raise RuntimeError(f"rank {rank}: HIP device {local} is not visible (device_count = {torch.cuda.device_count()})")
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device.type == "cuda":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    return rank, (dist.get_world_size() if dist.is_initialized() else 1), device


class SyntheticPairs:
    """Seeded synthetic training pairs: image1 is image0 resampled through the homography a fronto-parallel plane
    at depth `z` induces under a random small relative pose, so the pose is recoverable from the pixels.
    Yields `data` dicts already on `device` (B pairs, fp32 images in [0,1], T_0to1 [B,4,4])."""

    def __init__(self, B, H, W, device, seed=0, rank=0, z=3.0, max_deg=15.0, max_t=0.5):
        self.B, self.H, self.W, self.device = B, H, W, torch.device(device)
        self.g = torch.Generator().manual_seed(1_000_003 * (rank + 1) + seed)
        self.z, self.max_deg, self.max_t = z, max_deg, max_t
        f = 0.8 * max(H, W)
        self.K = torch.tensor([[f, 0, (W - 1) / 2], [0, f, (H - 1) / 2], [0, 0, 1.0]])

    def _pose(self):
        B, g = self.B, self.g
        axis = nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
        ang = torch.rand(B, generator=g) * math.radians(self.max_deg)
        kx = torch.zeros(B, 3, 3)
        kx[:, 0, 1], kx[:, 0, 2], kx[:, 1, 0] = -axis[:, 2], axis[:, 1], axis[:, 2]
        kx[:, 1, 2], kx[:, 2, 0], kx[:, 2, 1] = -axis[:, 0], -axis[:, 1], axis[:, 0]
        s, c = torch.sin(ang)[:, None, None], torch.cos(ang)[:, None, None]
        R = torch.eye(3) + s * kx + (1 - c) * (kx @ kx)
        t = (torch.rand(B, 3, generator=g) * 2 - 1) * self.max_t
        T = torch.eye(4).repeat(B, 1, 1)
        T[:, :3, :3], T[:, :3, 3] = R, t
        return T

    def batch(self):
        B, H, W, g = self.B, self.H, self.W, self.g
        # band-limited texture: a coarse random field upsampled (3 channels)
        coarse = torch.rand(B, 3, H // 8 + 2, W // 8 + 2, generator=g)
        im0 = nn.functional.interpolate(coarse, size=(H, W), mode="bicubic", align_corners=True).clamp(0, 1)
        T = self._pose()
        # x1 ~ K (R + t n^T / z) K^-1 x0 with n = e_z: sample image0 at the back-warped pixel of every image1 pixel
        n = torch.tensor([0.0, 0.0, 1.0])
        Hm = self.K @ (T[:, :3, :3] + T[:, :3, 3:] @ n[None, None, :] / self.z) @ torch.linalg.inv(self.K)
        Hinv = torch.linalg.inv(Hm)
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        p1 = torch.stack([xs, ys, torch.ones_like(xs)], 0).reshape(1, 3, -1)
        p0 = Hinv @ p1
        p0 = p0[:, :2] / p0[:, 2:].clamp_min(1e-6)
        gx = p0[:, 0] / (W - 1) * 2 - 1
        gy = p0[:, 1] / (H - 1) * 2 - 1
        grid = torch.stack([gx, gy], -1).reshape(B, H, W, 2)
        im1 = nn.functional.grid_sample(im0, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        put = (lambda x: x.pin_memory().to(self.device, non_blocking=True)) if self.device.type == "cuda" else (lambda x: x)
        return {"image0": put(im0.contiguous()), "image1": put(im1.contiguous()), "T_0to1": put(T)}

    def __iter__(self):
        while True:
            yield self.batch()


# ----------------------------------------------------------------------------------------------------------------
# the step
# ----------------------------------------------------------------------------------------------------------------
class _Step(nn.Module):
    """forward(data) -> (R_loss, t_loss, loss): prediction AND loss inside the module DDP wraps (see the module docstring)"""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, data):
        data = dict(data)
        self.model(data)
        return self.model.loss_fn(data)


class Trainer:
    """Owns the replica, the optimiser, the schedule and the step counter of one rank."""

    def __init__(self, cfg, device=None, sample=None):
        self.cfg = cfg
        from .. import options
        options.apply_cfg(cfg)                     # declared kernel-selection options under cfg.HIP (options.py)
        self.rank, self.world, self.device = init_distributed(device)
        torch.manual_seed(0)                                    # pl.seed_everything(0), train.py:25: identical replicas
        cls = {"Regression": RegressionModel, "RegressionMultiFrame": RegressionMultiFrameModel}[cfg.MODEL]
        self.model = cls(cfg).to(self.device)
        self.channels_last = bool(getattr(cfg.TRAINING, "CHANNELS_LAST", False))
        if self.channels_last:
            self.model = self.model.to(memory_format=torch.channels_last)
        self.precision = str(getattr(cfg.TRAINING, "PRECISION", "bf16"))
        if self.precision not in ("bf16", "fp32"):
            raise ValueError(f"TRAINING.PRECISION must be 'bf16' or 'fp32', got {self.precision!r}")
        if sample is not None:
            self.materialise(sample)
        self.step_mod = self.opt = self.sched = None
        self.global_step, self.epoch = 0, 0
        # TRAINING.GRAPH_STEP: forward + loss + backward of a step replayed from ONE captured HIP graph (the step is ~1900 kernel
        # launches for ~33 ms of GPU time at batch 10: launch-bound).  Gradients then live in one flat buffer that is all-reduced
        # with a single RCCL call after the replay (no DDP wrapper: its hooks cannot run inside a replay), clipped and handed to Adam.
        # (On a CPU device the same flat-gradient / single-all-reduce step runs with eager launches: that is what the gloo tests drive.)
        self.graph_step = bool(getattr(cfg.TRAINING, "GRAPH_STEP", False)) and not self.channels_last
        self._flat = self._gstep = self._gkeys = None

    # lazy layers (the heads size their first linear layer from the feature volume) must exist before the optimiser and
    # the gradient buckets are built
    def materialise(self, sample):
        was = self.model.training
        self.model.train()
        with torch.no_grad(), self.autocast():
            self.model({k: v for k, v in sample.items()})
        self.model.train(was)                                   # (DDP broadcasts rank 0's parameters and buffers when it wraps)
        if self.channels_last:                                  # the lazily created layers too
            self.model = self.model.to(memory_format=torch.channels_last)

    def autocast(self):
        if self.precision == "bf16":
            return torch.autocast(self.device.type, dtype=torch.bfloat16)
        return contextlib.nullcontext()

    def build(self):
        if any(isinstance(p, nn.parameter.UninitializedParameter) for p in self.model.parameters()):
            raise RuntimeError("lazy layers are not materialised: pass a sample batch to Trainer(...) or call materialise()")
        self.opt, self.sched = self.model.configure_optimizers()
        step = _Step(self.model)
        if self.graph_step:
            self._flatten_grads()
            if self.world > 1:                                  # what DDP does when it wraps: replicas start from rank 0's state
                for t in list(self.model.parameters()) + list(self.model.buffers()):
                    dist.broadcast(t.data, 0)
        elif self.world > 1:
            mb = int(getattr(self.cfg.TRAINING, "DDP_BUCKET_MB", 64))
            # the heads' sticky NaN / Inf flag is per-rank state: DDP's buffer broadcast (rank 0 -> all, every forward) would clear a flag raised on
            # another rank between two reads; BatchNorm's running statistics keep being broadcast
            step._ddp_params_and_buffers_to_ignore = [n for n, _ in step.named_buffers() if n.endswith(".invalid") or n == "invalid"]
            step = nn.parallel.DistributedDataParallel(
                step, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                bucket_cap_mb=mb, gradient_as_bucket_view=True, broadcast_buffers=True)
        self.step_mod = step
        return self

    # ---- graph-step mode ----
    def _flatten_grads(self):
        """every parameter's .grad becomes a view into ONE fp32 buffer: one zero kernel, one all-reduce, one norm"""
        params = [p for p in self.model.parameters() if p.requires_grad]
        self._flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=self.device)
        o = 0
        for p in params:
            p.grad = self._flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def _fwd_bwd(self, data):
        self._flat.zero_()
        with self.autocast():
            R_loss, t_loss, loss = self.step_mod(data)
        loss = loss.float().sum()
        loss.backward()                                         # accumulates in place into the flat views
        return R_loss.detach(), t_loss.detach(), loss.detach()

    def _graph_train_step(self, data):
        from ..nets.graph import GraphCaptureError, GraphedCall
        keys = sorted(k for k, v in data.items() if isinstance(v, torch.Tensor))
        shapes = tuple((k, tuple(data[k].shape)) for k in keys)
        if self._gstep is None and self._gkeys is None and self.device.type != "cuda":
            self._gkeys = ()
        if self._gstep is None and self._gkeys is None:
            try:
                self._gstep = GraphedCall(lambda *ts: self._fwd_bwd(dict(zip(keys, ts))), [data[k] for k in keys], warmup=3, clone_outputs=True)
                self._gkeys = shapes
            except GraphCaptureError as e:
                import warnings
                warnings.warn(f"regression.train: {e}; forward / backward run eagerly from now on")
                self._gkeys = ()
        if self._gstep is not None and shapes == self._gkeys:
            losses = self._gstep(*[data[k] for k in keys])
        else:                                                   # another batch shape (or capture failed): same arithmetic, eager launches
            losses = self._fwd_bwd(data)
        if self.world > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(self._flat, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(self._flat)
                self._flat.div_(self.world)
        clip = float(self.cfg.TRAINING.GRAD_CLIP or 0.0)
        if clip > 0:                                            # clip_grad_norm_'s formula on the flat buffer, no host read
            total = torch.linalg.vector_norm(self._flat, dtype=torch.float64)       # binary64 accumulation: 20 M terms in one reduction
            self._flat.mul_((clip / (total + 1e-6)).clamp(max=1.0).float())
        self.opt.step()
        if self.sched is not None:
            self.sched.step()
        self.global_step += 1
        return losses

    @torch.no_grad()
    def sync_buffers(self):
        """graph-step mode has no DDP wrapper broadcasting BatchNorm statistics: average the floating-point buffers over the ranks
        (before validation / checkpoints)"""
        if self.world > 1 and self.graph_step:
            for b in self.model.buffers():
                if b.dtype.is_floating_point:
                    dist.all_reduce(b)
                    b.div_(self.world)

    def train_step(self, data):
        """model.py:84-97 + Lightning's optimiser step: zero, forward, loss, backward (gradient all-reduce overlapped),
        clip, Adam, StepLR.  Returns the three loss tensors (device, no sync)."""
        if self.step_mod is None:
            self.build()
        self.model.train()
        if self.graph_step:
            return self._graph_train_step(data)
        self.opt.zero_grad(set_to_none=True)
        with self.autocast():
            R_loss, t_loss, loss = self.step_mod(data)
        loss = loss.float().sum()
        loss.backward()
        clip = float(self.cfg.TRAINING.GRAD_CLIP or 0.0)
        if clip > 0:
            nn.utils.clip_grad_norm_(self.model.parameters(), clip)
        self.opt.step()
        if self.sched is not None:
            self.sched.step()
        self.global_step += 1
        return R_loss.detach(), t_loss.detach(), loss.detach()

    @torch.no_grad()
    def validate(self, batches):
        """validation_step / on_validation_epoch_end: `batches` is THIS rank's share; the summary is computed from all ranks'"""
        self.sync_buffers()
        self.model.eval()
        outs = []
        for data in batches:
            with self.autocast():
                o = self.model.validation_outputs(dict(data))
            outs.append({k: v.detach().float().cpu() for k, v in o.items()})
        if self.world > 1:
            allo = [None] * self.world
            dist.all_gather_object(allo, outs)
            outs = [o for part in allo for o in part]
        self.model.train()
        return self.model.aggregate_validation(outs) if outs else {}

    # ---- checkpoints (Lightning layout) ----
    def state(self):
        return {"state_dict": self.model.state_dict(), "optimizer_states": [self.opt.state_dict()] if self.opt else [],
                "lr_schedulers": [self.sched.state_dict()] if self.sched else [], "epoch": self.epoch, "global_step": self.global_step}

    def save(self, path):
        self.sync_buffers()
        if self.rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            tmp = f"{path}.tmp{os.getpid()}"
            torch.save(self.state(), tmp)
            os.replace(tmp, path)
        if self.world > 1:
            dist.barrier()

    def resume(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        self.model.load_state_dict(ck["state_dict"])
        if self.step_mod is None:
            self.build()
        if ck.get("optimizer_states"):
            self.opt.load_state_dict(ck["optimizer_states"][0])
        if self.sched is not None and ck.get("lr_schedulers"):
            self.sched.load_state_dict(ck["lr_schedulers"][0])
        self.epoch, self.global_step = int(ck.get("epoch", 0)), int(ck.get("global_step", 0))

    def check_finite(self, loss):
        """Stop the run -- on EVERY rank, before the next optimizer step can be checkpointed -- when a head flagged NaN / Inf anchors
        (the reference's check, lib/models/regression/head.py:88-101, is a per-step sys.exit; here the heads raise a device flag and
        it is read at the log cadence together with the loss that is read anyway) or the loss is not finite."""
        bad = torch.zeros(1, device=self.device)
        flag = getattr(getattr(self.model, "head", None), "invalid", None)
        if flag is not None:
            bad += flag.to(bad.dtype).reshape(-1)[0]
            self.model.head.clear_invalid()                      # sticky between reads (head._flag ORs every forward in)
        bad += (~torch.isfinite(loss.detach().float().sum())).to(bad.dtype)
        if self.world > 1 and dist.is_initialized():
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if float(bad.item()) > 0:
            if self.rank == 0:
                print("Invalid anchors / non-finite loss!")
            raise SystemExit("Stopped")

    def fit(self, train_iter, steps_per_epoch, val_batches=None, out_dir=None, log=print):
        """epochs x steps with the reference's cadence: LOG_INTERVAL, validation at VAL_INTERVAL (fraction of an epoch
        or a step count), `last.ckpt` after every validation and `e{epoch}-last.ckpt` at every epoch end"""
        tcfg = self.cfg.TRAINING
        if self.step_mod is None:
            self.build()
        vi = tcfg.VAL_INTERVAL or 1.0
        val_every = max(1, int(round(vi * steps_per_epoch))) if vi <= 1 else int(vi)
        log_every = int(tcfg.LOG_INTERVAL or 50)
        it = iter(train_iter)
        last = {}
        start_epoch = self.epoch
        for self.epoch in range(start_epoch, int(tcfg.EPOCHS or 1)):
            for s in range(steps_per_epoch):
                t0 = time.perf_counter()
                R_loss, t_loss, loss = self.train_step(next(it))
                if self.global_step % log_every == 0:
                    self.check_finite(loss)                     # the reference stops on NaN / Inf (head.py:88-101 sys.exit('Stopped'))
                if self.global_step % log_every == 0 and self.rank == 0:
                    log(f"epoch {self.epoch} step {self.global_step}: loss {loss.item():.5f} R {R_loss.float().sum().item():.5f} "
                        f"t {t_loss.float().sum().item():.5f} ({1e3 * (time.perf_counter() - t0):.1f} ms)")
                if val_batches is not None and (s + 1) % val_every == 0:
                    last = self.validate(val_batches)
                    if self.rank == 0:
                        log(f"validation @ {self.global_step}: " + ", ".join(f"{k} {v:.4f}" for k, v in sorted(last.items())[:6]))
                    if out_dir:
                        self.check_finite(loss)                 # never checkpoint weights that went through a NaN step
                        self.save(os.path.join(out_dir, "last.ckpt"))
            if out_dir:
                self.check_finite(loss)
                self.epoch += 1                                 # a resumed run starts at the next epoch
                self.save(os.path.join(out_dir, f"e{self.epoch - 1}-last.ckpt"))
                self.epoch -= 1
        self.epoch = int(tcfg.EPOCHS or 1)
        return last


class _Reiterable:
    """an iterable that restarts from a factory each time it is iterated (the validation share of a rank)"""

    def __init__(self, factory):
        self.factory = factory

    def __iter__(self):
        return iter(self.factory())


def main(argv=None):
    """python -m mapfree_reloc_amd.regression.train <model.yaml> [<dataset.yaml>] [--synthetic B H W --steps-per-epoch K]
    (under torch.distributed.run for more than one GPU).  Without --synthetic the Map-free train / val splits under
    DATASET.DATA_ROOT are read (datasets.make_train_loaders)."""
    from ..config import get_cfg_defaults
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("dataset_config", nargs="?", default="")
    ap.add_argument("--experiment", default="default")
    ap.add_argument("--resume", default="")
    ap.add_argument("--synthetic", type=int, nargs=3, metavar=("B", "H", "W"), default=None,
                    help="train on seeded synthetic pairs (no dataset offline); without it a data root is required")
    ap.add_argument("--steps-per-epoch", type=int, default=100)
    ap.add_argument("--val-batches", type=int, default=4)
    a = ap.parse_args(argv)
    cfg = get_cfg_defaults()
    if a.dataset_config:
        cfg.merge_from_file(a.dataset_config)
    cfg.merge_from_file(a.config)
    rank, world, device = init_distributed()
    if a.synthetic is None:
        # the Map-free training split (DATASET.DATA_ROOT/train with overlaps.npz, scene-balanced sampling) and the val split
        import itertools
        from ..datasets import make_train_loaders
        tl, vl = make_train_loaders(cfg, device, rank, world)          # raises MissingDataError without a data root
        first = next(iter(tl))
        tr = Trainer(cfg, device, sample=first)
        tr.build()
        if a.resume:
            tr.resume(a.resume)
        nval = int(cfg.TRAINING.VAL_BATCHES or 0)
        val = _Reiterable(lambda: itertools.islice(iter(vl), max(1, nval // world) if nval else None))
        # every rank must run the SAME number of steps per epoch (DDP collectives, the sharded validation gather): agree on the minimum
        spe = torch.tensor([len(tl)], device=device)
        if world > 1:
            dist.all_reduce(spe, op=dist.ReduceOp.MIN)
        res = tr.fit(tl.forever(), int(spe.item()), val, out_dir=os.path.join("weights", a.experiment))
    else:
        B, H, W = a.synthetic
        src = SyntheticPairs(B, H, W, device, seed=0, rank=rank)
        tr = Trainer(cfg, device, sample=src.batch())
        tr.build()
        if a.resume:
            tr.resume(a.resume)
        val_src = SyntheticPairs(B, H, W, device, seed=10_007, rank=rank)
        val = [val_src.batch() for _ in range(max(1, a.val_batches // world))]
        res = tr.fit(src, a.steps_per_epoch, val, out_dir=os.path.join("weights", a.experiment))
    if rank == 0:
        print({k: round(v, 5) for k, v in res.items()})
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
