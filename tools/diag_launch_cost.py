"""host-side cost of issuing work on this box: ctypes kernel launches, torch GEMMs, allocations (no sync inside the loops)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mapfree_reloc_amd as m
lib = m._lib.load(require_gpu=True)
dev = "cuda"
d = torch.rand(1, 64, 64, device=dev); out = torch.empty(1, 16, device=dev)
sp = m._lib.stream_ptr()
def t(name, fn, n=2000):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / n * 1e6
    print(f"{name:42s} host issue {host:7.1f} us   incl. drain {tot:7.1f} us")
t("ctypes: mfr_depth_min (tiny kernel)", lambda: lib.mfr_depth_min(d.data_ptr(), 1, 64, 64, out.data_ptr(), sp))
t("ctypes + stream_ptr() lookup", lambda: lib.mfr_depth_min(d.data_ptr(), 1, 64, 64, out.data_ptr(), m._lib.stream_ptr()))
a = torch.rand(2048, 256, device=dev); w = torch.rand(768, 256, device=dev); bias = torch.rand(768, device=dev)
t("torch.addmm 2048x256x768", lambda: torch.addmm(bias, a, w.t()))
a2 = torch.rand(65536, 256, device=dev)
t("torch.addmm 65536x256x768", lambda: torch.addmm(bias, a2, w.t()), 300)
t("torch.empty(2,64,720,540)", lambda: torch.empty(2, 64, 720, 540, device=dev))
x = torch.rand(2, 64, 360, 270, device=dev)
t("torch elementwise add", lambda: x.add_(1.0))
