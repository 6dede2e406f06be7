"""why does the un-synchronised loop run slower? (diagnostic)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mapfree_reloc_amd import images as IM
from mapfree_reloc_amd.pipeline import SuperGluePnPPipeline
dev = torch.device("cuda:0")
B = 16
sb = IM.synthetic_batch(list(range(B)))
d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sb.items()}
pipe = SuperGluePnPPipeline(dev)
for _ in range(3):
    pipe(d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"])
torch.cuda.synchronize()
def run(mode, steps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); keep = []
    for i in range(steps):
        out = pipe(d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"])
        if mode == "sync": torch.cuda.synchronize()
        if mode == "keep": keep.append(out)
        if mode == "event":
            e = torch.cuda.Event(); e.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    ms = torch.cuda.memory_stats()
    print(f"{mode:8s} {dt:6.1f} ms/step  reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB  allocs {ms['num_device_alloc']} retries {ms['num_alloc_retries']}", flush=True)
for mode in ["nosync", "sync", "nosync", "keep", "event", "sync", "nosync"]:
    run(mode)
