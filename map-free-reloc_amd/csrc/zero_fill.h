// zero_fill.h -- "memset to zero" as a KERNEL node.
//
// Every entry point of this library must replay correctly from a captured HIP graph (nets/graph.py).  hipMemsetAsync is
// captured as a memset node; on this stack (ROCm 7.2) replays of graphs holding our memset nodes stopped re-zeroing the NMS
// candidate counters once the inputs changed (tools/diag_graph_phase.py: first replay right, later replays wrong or faulting),
// while kernel nodes replay faithfully.  So counters and masks are cleared by this kernel instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

static __global__ void __launch_bounds__(256) mfr_zero_words_kernel(uint32_t *__restrict__ p, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

static __global__ void __launch_bounds__(256) mfr_zero_bytes_kernel(uint8_t *__restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}

// zero `bytes` bytes at p on `stream`; returns hipSuccess or the launch error
static inline hipError_t mfr_zero_async(void *p, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    if ((((size_t)p | bytes) & 3) == 0) {
        const size_t n = bytes / 4;
        const unsigned g = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
        hipLaunchKernelGGL(mfr_zero_words_kernel, dim3(g), dim3(256), 0, stream, (uint32_t *)p, n);
    } else {
        const unsigned g = (unsigned)((bytes + 255) / 256 > 4096 ? 4096 : (bytes + 255) / 256);
        hipLaunchKernelGGL(mfr_zero_bytes_kernel, dim3(g), dim3(256), 0, stream, (uint8_t *)p, bytes);
    }
    return hipGetLastError();
}
