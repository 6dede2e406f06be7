"""A/B of the fine-level encoder layer's MLP tail (LoFTR d_model 128) at the bench step's row counts: two launches (mfr_gemm_* 256 -> 256 + ReLU, then
mfr_gemm_*_ln 256 -> 128 + LayerNorm + residual) against the fused kernel (mfr_mlp_ln_*); plus the HBM traffic each form needs.
python tools/bench_fine_mlp.py [out.json] [rows ...]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mapfree_reloc_amd  # noqa: F401
from mapfree_reloc_amd.nets.linear import FusedMlpLn, SplitLinear

dev = "cuda:0"
rows = [int(v) for v in sys.argv[2:]] or [2_400_000, 1_200_000, 200_000]


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator().manual_seed(0)
w1 = (torch.randn(256, 256, generator=g) / 16).to(dev); w2 = (torch.randn(128, 256, generator=g) / 16).to(dev)
gam, bet = torch.ones(128, device=dev), torch.zeros(128, device=dev)
l1, l2 = SplitLinear(w1), SplitLinear(w2)
mlp = FusedMlpLn(l1, l2)
res = {}
for M in rows:
    xm = torch.randn(M, 256, device=dev)
    hid = torch.empty(M, 256, device=dev)
    t1 = timed(lambda: l1(xm, out=hid, relu=True))
    t2 = timed(lambda: l2(hid, out=xm[:, :128], ln=(gam, bet), accumulate=True))
    tf = timed(lambda: mlp(xm, out=xm[:, :128], ln=(gam, bet), accumulate=True))
    U = M * 128 * 4 / 1e9
    flops = 3 * 2.0 * M * (256 * 256 + 256 * 128)
    res[str(M)] = dict(rows=M, l1_relu_ms=round(t1, 4), l2_ln_res_ms=round(t2, 4), two_launch_ms=round(t1 + t2, 4), fused_ms=round(tf, 4),
                       hbm_GB=dict(two_launch=round(8 * U, 3), fused=round(3 * U, 3)), fused_hbm_TBs=round(3 * U / tf, 2), two_launch_hbm_TBs=round(8 * U / (t1 + t2), 2),
                       fused_executed_tflops=round(flops / tf / 1e9, 1), fused_mfma_pipe_frac=round(flops / tf / 1e9 / 2500, 4))
    print(json.dumps(res[str(M)]), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
