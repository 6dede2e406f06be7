"""Where does the HIP-graph path of the whole SuperGlue+PnP step fault?  Prints (flushed) after every phase:
A eager on the default stream, B eager on a side stream, C capture, D replays.  usage: diag_graph_phase.py [pairs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mapfree_reloc_amd import images as IM
from mapfree_reloc_amd.pipeline import SuperGluePnPPipeline


def say(*a):
    print(*a, flush=True)


dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in IM.synthetic_batch(list(range(B))).items()}
d2 = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in IM.synthetic_batch(list(range(100, 100 + B))).items()}
pipe = SuperGluePnPPipeline(dev)
keys = ("images", "depth0", "K0", "K1", "pair_ids")
run = lambda x: pipe(*[x[k] for k in keys])
for i in range(2):
    ref = run(d); torch.cuda.synchronize()
say("A ok: eager, default stream")
static = {k: d[k].clone() for k in keys}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        out = run(static); s.synchronize()
        say(f"B{i} ok: eager, side stream; equal to default-stream result: {torch.equal(out['n_inliers'], ref['n_inliers'])}")
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gout = run(static)
torch.cuda.synchronize()
say("C ok: captured")
for i in range(4):
    for k in keys:
        static[k].copy_((d2 if i & 1 else d)[k])
    g.replay(); torch.cuda.synchronize()
    e = run(d2 if i & 1 else d); torch.cuda.synchronize()
    say(f"D{i} ok: replay; equal to eager: {torch.equal(gout['n_inliers'], e['n_inliers'])} {torch.equal(gout['R'], e['R']) or bool((torch.isnan(gout['R']) == torch.isnan(e['R'])).all())}")
say("done")
