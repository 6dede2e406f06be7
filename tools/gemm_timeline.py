"""In-kernel timeline of the operand-splitting linear-layer kernel (csrc/gemm_split.hip, the default persistent kernel): s_memtime stamps of the
four wavefronts of one mid-grid workgroup over its first 12 K steps, from a MEASUREMENT build of the same source (-DGD_PROF ->
tools/ubench/libgemm_prof.so; `python tools/gemm_timeline.py --build` on the CPU box).  python tools/gemm_timeline.py [--shape mlp1|qkv|loftr] [out.json]"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "ubench", "libgemm_prof.so")
SRC = os.path.join(ROOT, "map-free-reloc_amd", "csrc", "gemm_split.hip")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", "-DGD_PROF",
                           "-I" + os.path.join(ROOT, "include"), SRC, "-o", SO])
    print("built", SO)
    sys.exit(0)
import torch  # noqa: E402
arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
shape = arg("--shape", "mlp1")
M, K, N, flags = {"mlp1": (65536, 512, 512, 1), "qkv": (65536, 256, 768, 0), "loftr": (195840, 256, 256, 0)}[shape]
lib = C.CDLL(SO)
vp, i = C.c_void_p, C.c_int
lib.mfr_gemm_f16x2_pack_bytes.restype = C.c_size_t
dev = "cuda:0"
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
pk = torch.empty(lib.mfr_gemm_f16x2_pack_bytes(N, K), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert lib.mfr_gemm_f16x2_pack(vp(w.data_ptr()), i(N), i(K), vp(pk.data_ptr()), vp(st)) == 0
run = lambda: lib.mfr_gemm_f16x2(vp(x.data_ptr()), i(K), vp(pk.data_ptr()), vp(b.data_ptr()), vp(y.data_ptr()), i(N), i(M), i(N), i(K), i(flags), vp(st))
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert lib.mfr_gemm_split_profile(buf) == 0
waves = [[int(buf[wv * 64 + k]) for k in range(64)] for wv in range(4)]
t0 = min(t[0] for t in waves)
what = ["step top", "barrier 1 passed (previous fragments read)", "X split + stored, W DMA + X loads issued", "vmcnt: this step's W landed", "barrier 2 passed"]
rows = []
print(json.dumps({"shape": shape, "launch_ms_instrumented": round(e0.elapsed_time(e1) / 10, 4), "k_steps_per_tile": K // 32}))
prev = 0
for s in range(12):
    for k in range(5):
        v = [t[5 * s + k] - t0 for t in waves]
        rows.append({"step": s, "stamp": k, "what": what[k], "min": min(v), "max": max(v), "per_wave": v})
        print(f"step {s:2d} {what[k]:46s} min {min(v):7d} max {max(v):7d} (+{max(v) - prev:6d})")
        prev = max(v)
if sys.argv[-1].endswith(".json"):
    json.dump({"shape": shape, "launch_ms_instrumented": round(e0.elapsed_time(e1) / 10, 4), "timeline": rows}, open(sys.argv[-1], "w"), indent=1)
