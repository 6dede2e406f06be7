"""RegressionModel: encoder(image0), encoder(image1) -> aggregator -> head -> (R, t) written into `data`
(reference: lib/models/regression/model.py:14-97,230-248; builder lib/models/builder.py:8-27).

The reference wraps this in a pytorch_lightning module; the training loop here is regression/train.py (one process per
GPU, DDP over RCCL).  State-dict keys (`encoder.*`, `aggregator.*`, `head.*`, `s_r`, `s_t`) are the reference's, and
`load_checkpoint` reads a Lightning checkpoint's `state_dict`."""
import torch
import torch.nn as nn

from .aggregator import AGGREGATORS
from .encoder import ENCODERS
from .head import HEADS
from .losses import LOSSES, A_metrics, error_auc, pose_error_torch


def _pick(table, name, what):
    try:
        return table[name]
    except KeyError:
        raise NotImplementedError(f"Invalid {what} {name}") from None


class RegressionModel(nn.Module):
    """Regresses the relative pose between a pair of images"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.encoder = _pick(ENCODERS, cfg.ENCODER.TYPE, "encoder")(cfg.ENCODER)
        self.aggregator = _pick(AGGREGATORS, cfg.AGGREGATOR.TYPE, "aggregator")(cfg.AGGREGATOR, self.encoder.num_out_layers)
        self.head = _pick(HEADS, cfg.HEAD.TYPE, "head")(cfg, self.aggregator.num_out_layers)
        self.rot_loss = _pick(LOSSES, cfg.TRAINING.ROT_LOSS, "rotation loss")
        self.trans_loss = _pick(LOSSES, cfg.TRAINING.TRANS_LOSS, "translation loss")
        self.LAMBDA = cfg.TRAINING.LAMBDA
        if self.LAMBDA == 0.:                       # learnable loss weighting (Kendall & Cipolla)
            self.s_r = nn.Parameter(torch.zeros(1))
            self.s_t = nn.Parameter(torch.zeros(1))

    def _second_image(self, data):
        return data["image1"]

    def forward(self, data):
        im0, im1 = data["image0"], self._second_image(data)
        # one encoder pass over both images of every pair: same weights, half the launches, convolutions at twice the batch.  The
        # BatchNorm layers keep the reference's arithmetic (two encoder calls, model.py:64-66): statistics per view, running estimates
        # updated view 0 first (encoder.ViewBatchNorm2d) -- same outputs, gradients and buffers as the two-call path
        if self.training and getattr(self.cfg.TRAINING, "SIAMESE_BATCH", False) and im0.shape == im1.shape:
            from .encoder import view_groups
            with view_groups(2):
                vol = self.encoder(torch.cat([im0, im1], 0))
            vol0, vol1 = vol[:im0.shape[0]], vol[im0.shape[0]:]
        else:
            vol0, vol1 = self.encoder(im0), self.encoder(im1)
        R, t = self.head(self.aggregator(vol0, vol1), data)
        data["R"], data["t"], data["inliers"] = R, t, 0
        if not self.training:
            self.check_finite()
        return R, t

    def check_finite(self):
        """raise the reference's exit (head.py:88-101) if a head flagged NaN/Inf outputs; one host read"""
        flag = getattr(self.head, "invalid", None)
        if flag is not None:
            bad = bool(flag)
            self.head.clear_invalid()
            if bad:
                print("Invalid anchors!")
                raise SystemExit("Stopped")

    def loss_fn(self, data):
        R_loss, t_loss = self.rot_loss(data), self.trans_loss(data)
        if self.LAMBDA == 0:
            loss = R_loss * torch.exp(-self.s_r) + t_loss * torch.exp(-self.s_t) + self.s_r + self.s_t
        else:
            loss = R_loss + self.LAMBDA * t_loss
        return R_loss, t_loss, loss

    @torch.no_grad()
    def validation_outputs(self, data):
        R, t = self(data)
        out = pose_error_torch(R, t, data["T_0to1"], reduce=None)
        out["R_loss"], out["t_loss"], out["loss"] = (x.reshape(1) for x in self.loss_fn(data))
        return out

    @staticmethod
    def aggregate_validation(outputs):
        """the logged validation summary (model.py:116-186) from a list of validation_outputs()"""
        cat = {k: torch.cat([o[k].reshape(-1).float().cpu() for o in outputs]) for k in outputs[0]}
        res = {"val_loss/R_loss": cat["R_loss"].mean().item(), "val_loss/t_loss": cat["t_loss"].mean().item(),
               "val_loss/loss": cat["loss"].mean().item(), "val_metrics/t_ang_err": cat["t_err_ang"].median().item(),
               "val_metrics/t_scale_err": cat["t_err_scale"].median().item(),
               "val_metrics/t_euclidean_err": cat["t_err_euc"].median().item(), "val_metrics/R_err": cat["R_err"].median().item()}
        pose = torch.maximum(cat["t_err_ang"], cat["R_err"]).numpy()
        for name, err, thr in (("euc", cat["t_err_euc"].numpy(), (0.1, 0.5, 1.0)), ("pose", pose, (5, 10, 20)),
                               ("rot", cat["R_err"].numpy(), (5, 10, 20)), ("tang", cat["t_err_ang"].numpy(), (5, 10, 20))):
            for th, v in zip(thr, error_auc(err, thr).values()):
                res[f"val_auc/{name}_{int(th * 100) if name == 'euc' else th}"] = v
        for k, v in zip((1, 2, 3), A_metrics(cat["t_err_scale_sym"])):
            res[f"val_t_scale/a{k}"] = float(v)
        return res

    def configure_optimizers(self):
        tcfg = self.cfg.TRAINING
        fused = next(self.parameters()).is_cuda
        opt = torch.optim.Adam(self.parameters(), lr=tcfg.LR, eps=1e-6, fused=fused)
        sched = torch.optim.lr_scheduler.StepLR(opt, tcfg.LR_STEP_INTERVAL, tcfg.LR_STEP_GAMMA) if tcfg.LR_STEP_INTERVAL else None
        return opt, sched

    def load_checkpoint(self, path, strict=True):
        """a Lightning checkpoint ({'state_dict': ...}) or a bare state dict"""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        return self.load_state_dict(ckpt.get("state_dict", ckpt), strict=strict)


class RegressionMultiFrameModel(RegressionModel):
    """multi-frame queries: image1 is [B, T, 3, H, W]; the last frame is the one regressed against (model.py:236-248)"""

    def _second_image(self, data):
        return data["image1"][:, -1]


def build_regression_model(cfg, checkpoint=""):
    cls = {"Regression": RegressionModel, "RegressionMultiFrame": RegressionMultiFrameModel}[cfg.MODEL]
    model = cls(cfg)
    if checkpoint:
        model.load_checkpoint(checkpoint)
    if torch.cuda.is_available():
        model = model.cuda()
    return model.eval()
