"""In-kernel timeline of the operand-splitting Winograd kernel (csrc/winograd_split.hip) on one layer: s_memtime stamps of the eight wavefronts
of one mid-grid workgroup, from a MEASUREMENT build of the same source (-DWB_PROF -> tools/ubench/libwino_prof.so; build it on the CPU box:
`python tools/conv_timeline.py --build`), so the product library carries no instrumentation.
    python tools/conv_timeline.py [--layer conv1ab|conv1b|conv2a|l1out2] [--split f16x2|bf16x3] [out.json]"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "ubench", "libwino_prof.so")
SRC = os.path.join(ROOT, "map-free-reloc_amd", "csrc", "winograd_split.hip")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", "-fno-slp-vectorize", "-DWB_PROF",
                           "-I" + os.path.join(ROOT, "include"), SRC, "-o", SO])
    print("built", SO)
    sys.exit(0)
import torch  # noqa: E402
arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
layer, split = arg("--layer", "conv1b"), arg("--split", "f16x2")
B, ci, co, H, W, pool, act = {"conv1ab": (64, 64, 64, 720, 540, 1, 1), "conv1b": (64, 64, 64, 720, 540, 1, 1), "conv2a": (64, 64, 64, 360, 270, 0, 1), "l1out2": (32, 196, 196, 360, 272, 0, 2)}[layer]
lib = C.CDLL(SO)
vp, i = C.c_void_p, C.c_int
getattr(lib, f"mfr_wino_{split}_filter_bytes").restype = C.c_size_t
dev = "cuda:0"
x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) / (3.0 * ci ** 0.5); b = torch.randn(co, device=dev)
u = torch.empty(getattr(lib, f"mfr_wino_{split}_filter_bytes")(ci, co), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert getattr(lib, f"mfr_wino_{split}_filter_transform")(vp(w.data_ptr()), i(ci), i(co), vp(u.data_ptr()), vp(st)) == 0
y = torch.empty((B, co, H // 2, W // 2) if pool else (B, co, H, W), device=dev)
conv = getattr(lib, f"mfr_conv3x3_wino_{split}")
run = lambda: conv(vp(x.data_ptr()), vp(u.data_ptr()), vp(b.data_ptr()), None, i(B), i(ci), i(co), i(H), i(W), i(act), i(pool), vp(y.data_ptr()), vp(st))
if layer == "conv1ab":          # SuperPoint's fused first two layers: stamps 1 = gray window in LDS, 2 = this wavefront's conv1a channels done, 3 = patch barrier passed
    gray = torch.rand(B, 1, H, W, device=dev); w1 = torch.randn(64, 1, 3, 3, device=dev) / 3.0; b1 = torch.randn(64, device=dev) * 0.3
    run = lambda: lib.mfr_sp_conv1ab_f16x2(vp(gray.data_ptr()), vp(w1.data_ptr()), vp(b1.data_ptr()), vp(u.data_ptr()), vp(b.data_ptr()), i(B), i(H), i(W), vp(y.data_ptr()), vp(st))
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert lib.mfr_wino_split_profile(buf) == 0
names = {0: "start", 1: "filters + patches of step 0 arrived", 2: "prologue barrier passed", 3: "first V ready / loop top", 16: "loop exit", 17: "last step done",
         18: "round 0: barrier (stages dead)", 19: "round 0: partials written", 20: "round 0: barrier", 21: "round 0: transformed", 22: "round 1: barrier", 23: "round 1: partials written",
         24: "round 1: barrier", 25: "round 1: transformed", 26: "end (stores issued)"}
for c in range(4):
    names[4 + 3 * c] = f"step {c}: 3 product blocks done"; names[5 + 3 * c] = f"step {c}: next patches in + fixed"; names[6 + 3 * c] = f"step {c}: barrier passed"
nks = (ci + 15) // 16
waves = [[int(buf[wv * 32 + k]) for k in range(32)] for wv in range(8)]
t0 = min(t[0] for t in waves)
rows = []
order = [0, 1, 2, 3] + [k for c in range(min(nks - 1, 4)) for k in (4 + 3 * c, 5 + 3 * c, 6 + 3 * c)] + [16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26]
for k in order:
    v = [t[k] - t0 for t in waves]
    rows.append({"stamp": k, "what": names.get(k, ""), "min": min(v), "max": max(v), "per_wave": v})
out = {"layer": layer, "split": split, "launch_ms_instrumented": round(e0.elapsed_time(e1) / 5, 4), "k_steps": nks,
       "note": "s_memtime ticks (100 MHz constant clock on gfx950? see ratio below) since the workgroup's first stamp; K steps beyond the 4th of a layer overwrite stamps 4..15 (c & 3)",
       "timeline": rows}
print(json.dumps({k: v for k, v in out.items() if k != "timeline"}))
prev = 0
for r in rows:
    print(f'{r["stamp"]:3d} {r["what"]:42s} min {r["min"]:8d} max {r["max"]:8d}  (+{r["max"] - prev:7d})')
    prev = r["max"]
if sys.argv[-1].endswith(".json"):
    json.dump(out, open(sys.argv[-1], "w"), indent=1)
