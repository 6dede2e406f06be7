"""HIP-graph replay of a fixed-shape device function (torch.cuda.CUDAGraph = hipGraph on ROCm).

The per-pair plugin API runs the matcher at batch 1 (lib/models/matching/model.py:30 asserts it), where a SuperPoint +
SuperGlue forward is ~300 kernel launches of a few microseconds each: issued eagerly the pair is host-bound (~40 ms); replayed
from one captured graph it costs the GPU time only (~4 ms; tools/diag_graph.py).  Everything inside the captured function must
be free of host synchronisation and of library calls that stage arguments through host memory.

Measured limit (ROCm 7.2 / PyTorch 2.10, tools/diag_graph.py): captures with >= 4 pairs fault at replay -- the large-M GEMMs
select hipBLASLt "UserArgs" kernels whose argument buffers do not survive capture -- so only batches of 1-2 pairs are graphed.
In a fresh process the batch-1 replay is exact and runs the matcher in 4-5 ms per pair; in processes that had already run other
GPU work (the solver plugins, a second matcher instance) the same capture faulted at replay twice, so the switch
(cfg.HIP.GRAPH_BATCH1) is OFF by default until the library-side cause is understood.
"""
import torch

MAX_GRAPH_PAIRS = 2


class GraphedCall:
    def __init__(self, fn, example_inputs, warmup=2):
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the capture stream: lazy initialisation, workspaces
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.static_out
