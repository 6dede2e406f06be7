// times the stages of the LDS 5-point solver (csrc/emat_lds.h): build with -DFP_STAGE_LIMIT=k, k = 0..5, 99
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../include -I../../map-free-reloc_amd/csrc -DFP_STAGE_LIMIT=k -o fivept_k fivept_time.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "emat_lds.h"
using namespace mfr;
__global__ void __launch_bounds__(64) k(const double *x0, const double *x1, double *Es, int *nsol)
{
    __shared__ double fp_lds[FP_LDS_DOUBLES * 64];
    __shared__ int fp_colp[9 * 64];
    const int it = blockIdx.x * 64 + threadIdx.x;
    double a[10], c[10];
    for (int q = 0; q < 10; ++q) { a[q] = x0[it * 10 + q]; c[q] = x1[it * 10 + q]; }
    nsol[it] = fivept_lds(a, c, Es + (size_t)it * 90, fp_lds + threadIdx.x, fp_colp + threadIdx.x);
}
int main()
{
    const int n = 250 * 64;
    std::mt19937 g(1); std::normal_distribution<double> nd(0, 0.3);
    std::vector<double> h0(n * 10), h1(n * 10);
    for (int i = 0; i < n; ++i) {                           // 5 points of a random two-view geometry: x1 ~ x0 + parallax
        for (int q = 0; q < 5; ++q) { double X = nd(g), Y = nd(g), Z = 3 + nd(g); h0[i * 10 + 2 * q] = X / Z; h0[i * 10 + 2 * q + 1] = Y / Z;
            h1[i * 10 + 2 * q] = (X + 0.3) / (Z + 0.05); h1[i * 10 + 2 * q + 1] = (Y + 0.02) / (Z + 0.05); }
    }
    double *d0, *d1, *Es; int *ns;
    hipMalloc(&d0, n * 80); hipMalloc(&d1, n * 80); hipMalloc(&Es, (size_t)n * 720); hipMalloc(&ns, n * 4);
    hipMemcpy(d0, h0.data(), n * 80, hipMemcpyHostToDevice); hipMemcpy(d1, h1.data(), n * 80, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<n / 64, 64>>>(d0, d1, Es, ns); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<n / 64, 64>>>(d0, d1, Es, ns); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<int> hn(n); hipMemcpy(hn.data(), ns, n * 4, hipMemcpyDeviceToHost);
    long tot = 0; for (int v : hn) tot += v;
    printf("FP_STAGE_LIMIT=%d: %.3f ms for %d solves (250 workgroups), mean solutions %.2f\n", FP_STAGE_LIMIT, ms, n, (double)tot / n);
    return 0;
}
