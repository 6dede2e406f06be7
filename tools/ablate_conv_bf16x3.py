"""timing ablations of the bf16x3 Winograd kernel on conv1b (64 images): which part of the K step / workgroup costs what"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapfree_reloc_amd import _lib
lib = _lib.load(require_gpu=True)
dev = "cuda:0"
n, ci, co, H, W = 64, 64, 64, 720, 540
if len(sys.argv) > 2: ci = co = int(sys.argv[2]); H, W = 180, 136
x = torch.randn(n, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) / 24; b = torch.randn(co, device=dev)
y = torch.empty(n, co, H // 2, W // 2, device=dev)
u = torch.empty(lib.mfr_wino_bf16x3_filter_bytes(ci, co), dtype=torch.uint8, device=dev)
lib.mfr_wino_bf16x3_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr())
res = {}
names = {0: "full", 1: "no epilogue", 2: "no filter loads", 4: "no V production", 8: "no patch staging", 3: "no epilogue, no filter loads", 7: "no epi / filters / V",
         15: "MFMA only", 14: "epilogue + MFMA only", 6: "no filter loads, no V", 12: "no V, no staging"}
for v in (0, 1, 2, 4, 8, 3, 6, 12, 7, 14, 15, 0):
    for _ in range(2):
        rc = lib.mfr_conv3x3_wino_bf16x3_variant(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), None, n, ci, co, H, W, 1, 1, v, _lib.ptr(y), _lib.stream_ptr())
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.mfr_conv3x3_wino_bf16x3_variant(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), None, n, ci, co, H, W, 1, 1, v, _lib.ptr(y), _lib.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    res[f"{v}: {names[v]}"] = round(e0.elapsed_time(e1) / 5, 3)
    print(v, names[v], res[f"{v}: {names[v]}"], flush=True)
# in-kernel timeline of one workgroup (s_memtime stamps, 100 MHz constant clock -> reported in ns)
import numpy as np
lib.mfr_conv3x3_wino_bf16x3_variant(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), None, n, ci, co, H, W, 1, 1, 16, _lib.ptr(y), _lib.stream_ptr())
torch.cuda.synchronize()
prof = np.zeros((4, 64), np.uint64)
assert lib.mfr_wino_bf16x3_profile(prof.ctypes.data) == 0
t = prof.astype(np.int64)
base = t[:, 0].min()
nks = (ci + 15) // 16
for w in range(4):
    row = t[w] - base
    names = [(0, "start"), (1, "stage0 in LDS"), (2, "w,V(0,0) made")]
    for c in range(nks - 1):
        names += [(3 + 6 * c, f"step{c} top"), (4 + 6 * c, "ph0 done"), (5 + 6 * c, "ph1 done"), (6 + 6 * c, "ph2 done"), (7 + 6 * c, "pstore/aload/pload issued"), (8 + 6 * c, "barrier passed")]
    names += [(3 + 6 * (nks - 1), "last step top"), (40, "loop end"), (41, "bias+barrier"), (42, "partials written"), (43, "barrier"), (44, "partials read"), (45, "end")]
    print("wave", w, " ".join(f"{nm}={int(row[k])}" for k, nm in names))
res["profile_ticks"] = (t - base).tolist()
if len(sys.argv) > 1: json.dump(res, open(sys.argv[1], "w"), indent=1)
