"""HIP-graph replay of a fixed-shape device function (torch.cuda.CUDAGraph = hipGraph on ROCm).

The per-pair plugin API runs the matcher at batch 1 (lib/models/matching/model.py:30 asserts it), where a SuperPoint +
SuperGlue forward is ~300 kernel launches of a few microseconds each: issued eagerly the pair is host-bound (~29 ms); replayed
from one captured graph it costs the GPU time only (~4 ms; tools/diag_graph.py).  Everything inside the captured function must
be free of host synchronisation.

What made replays unreliable in rounds 1-2 (first replay right, later replays wrong or faulting once the inputs changed,
tools/diag_graph_phase.py): the C-ABI cleared its counters / masks with hipMemsetAsync, which is captured as a MEMSET NODE, and
on this stack (ROCm 7.2) those nodes did not reliably re-zero the buffers on replay -- the NMS candidate counters kept growing
across replays.  The library now clears with a kernel (csrc/zero_fill.h); with kernel nodes only, captures of 1, 8 and 32 pairs
replay bit-identically to eager with changing inputs, with hipBLASLt GEMMs included.
"""
import torch


def _clone(x):
    if isinstance(x, torch.Tensor):
        return x.clone()
    if isinstance(x, dict):
        return {k: _clone(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_clone(v) for v in x)
    return x


class GraphCaptureError(RuntimeError):
    """capture of the function failed; the caller should run it eagerly (the process stays usable)"""


class GraphedCall:
    """fn(*tensors) captured once for the shapes of `example_inputs`; __call__ copies the new inputs into the static input
    buffers (device-to-device, part of the call) and replays.  clone_outputs: hand out copies instead of the static output
    buffers (needed when the caller keeps results across calls)."""

    def __init__(self, fn, example_inputs, warmup=2, clone_outputs=False):
        self.static_in = [t.clone() for t in example_inputs]
        self.clone_outputs = clone_outputs
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the capture stream: lazy initialisation, workspaces
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        before = torch.cuda.current_stream()
        try:
            # thread_local: other threads (a prefetching loader pinning memory / issuing H2D copies) must not invalidate the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.static_out = fn(*self.static_in)
        except Exception as e:
            # torch's context manager does not restore the stream when ending an invalidated capture raises: without this every
            # later launch of the process would go to the dead capture stream
            torch.cuda.set_stream(before)
            self.graph = None
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            raise GraphCaptureError(f"HIP-graph capture failed ({type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''})") from e

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            s.copy_(t, non_blocking=True)
        self.graph.replay()
        return _clone(self.static_out) if self.clone_outputs else self.static_out
