"""In-kernel timeline of the direct halo-staged convolution (csrc/conv_direct.hip) on one layer: s_memtime stamps of the eight wavefronts of one
mid-grid workgroup, from a MEASUREMENT build of the same source (-DDC_PROF -> tools/ubench/libdconv_prof.so; build it on the CPU box:
`python tools/dconv_timeline.py --build`), so the product library carries no instrumentation.
    python tools/dconv_timeline.py [--layer NAME] [out.json]"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "ubench", "libdconv_prof.so")
SRC = os.path.join(ROOT, "map-free-reloc_amd", "csrc", "conv_direct.hip")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", "-DDC_PROF",
                           "-I" + os.path.join(ROOT, "include"), SRC, "-o", SO])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", "-DDC_PROF", "-DDC_NW=8",
                           "-I" + os.path.join(ROOT, "include"), SRC, "-o", SO.replace(".so", "_nw8.so")])
    print("built", SO)
    sys.exit(0)
import torch  # noqa: E402
arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
layer = arg("--layer", "l1")
B, ci, co, H, W, pool, act, res = {"conv1b": (64, 64, 64, 540, 720, 1, 1, 0), "conv2a": (64, 64, 64, 270, 360, 0, 1, 0), "l1": (32, 128, 128, 272, 360, 0, 1, 1),
                                   "l1nores": (32, 128, 128, 272, 360, 0, 1, 0), "l1out2": (32, 196, 196, 272, 360, 0, 2, 0), "l1small": (1, 128, 128, 64, 360, 0, 1, 0), "l1tiny": (1, 128, 128, 16, 128, 0, 1, 0), "conv2asmall": (1, 64, 64, 64, 360, 0, 1, 0), "l2out2": (32, 256, 256, 136, 180, 0, 2, 0)}[layer]
lib = C.CDLL(SO.replace(".so", "_nw8.so") if "--nw8" in sys.argv else SO)
vp, i = C.c_void_p, C.c_int
lib.mfr_conv3x3_direct_f16x2_filter_bytes.restype = C.c_size_t
dev = "cuda:0"
x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) / (3.0 * ci ** 0.5); b = torch.randn(co, device=dev)
r = torch.randn(B, co, H, W, device=dev) if res else None
u = torch.empty(lib.mfr_conv3x3_direct_f16x2_filter_bytes(ci, co), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert lib.mfr_conv3x3_direct_f16x2_filter_pack(vp(w.data_ptr()), i(ci), i(co), vp(u.data_ptr()), vp(st)) == 0
y = torch.empty((B, co, H // 2, W // 2) if pool else (B, co, H, W), device=dev)
run = lambda: lib.mfr_conv3x3_direct_f16x2(vp(x.data_ptr()), vp(u.data_ptr()), vp(b.data_ptr()), vp(r.data_ptr()) if res else None, i(B), i(ci), i(co), i(H), i(W), i(act), i(pool), vp(y.data_ptr()), vp(st))
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
assert lib.mfr_dconv_profile(buf) == 0
nks = (ci + 15) // 16
names = {0: "start", 1: "stage 0 written, first weights requested", 34: "loop exit", 36: "scale / bias in registers", 37: "first channel block stored", 35: "end (stores issued)"}
for c in range(min(nks, 8)):
    names[2 + 4 * c] = f"step {c}: at the barrier"; names[3 + 4 * c] = f"step {c}: barrier passed, xl of tap 0 requested"
    names[4 + 4 * c] = f"step {c}: taps 0-2 done"; names[5 + 4 * c] = f"step {c}: taps 0-5 done"
waves = [[int(buf[wv * 64 + k]) for k in range(64)] for wv in range(8 if "--nw8" in sys.argv else 4)]
t0 = min(t[0] for t in waves)
order = [0, 1] + [k for c in range(min(nks, 8)) for k in (2 + 4 * c, 3 + 4 * c, 4 + 4 * c, 5 + 4 * c)] + [34, 36, 37, 35]
rows = []
for k in order:
    v = [t[k] - t0 for t in waves]
    rows.append({"stamp": k, "what": names.get(k, ""), "min": min(v), "max": max(v), "per_wave": v})
out = {"layer": layer, "launch_ms_instrumented": round(e0.elapsed_time(e1) / 5, 4), "k_steps": nks,
       "note": "s_memtime ticks since the workgroup's first stamp; K steps beyond the 8th overwrite stamps (c & 7)", "timeline": rows}
print(json.dumps({k: v for k, v in out.items() if k != "timeline"}))
prev = 0
for rr in rows:
    print(f'{rr["stamp"]:3d} {rr["what"]:48s} min {rr["min"]:8d} max {rr["max"]:8d}  (+{rr["max"] - prev:7d})')
    prev = rr["max"]
if sys.argv[-1].endswith(".json"):
    json.dump(out, open(sys.argv[-1], "w"), indent=1)
