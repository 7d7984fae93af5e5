// Micro-benchmark of the wave-level factor/solve (wave_linalg.h) in isolation: each wave finishes G = 64/DP random
// SPD systems.  Prints time per launch for several wave counts, so that throughput and latency can be separated.
#include "../bayesiandatafusion.jl_amd/csrc/wave_linalg.h"
#include <cstdio>
#include <cstdlib>
#ifndef WAVES_PER_EU
#define WAVES_PER_EU 2
#endif
void bdf_set_error(const char *, ...) {}

template <int DP>
__global__ __launch_bounds__(256, WAVES_PER_EU) void k_finish(double *out, int nwaves, const uint32_t *sweep)
{
    constexpr int LD = DP + 1;
    __shared__ __attribute__((aligned(16))) double lds[4 * (DP * LD + 128)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= nwaves) return;
    double *wl = lds + wave * (DP * LD + 128);
    double *tri = wl;
    const int c = lane % DP;
    double col[DP];
#pragma unroll
    for (int i = 0; i < DP; i++) {
        const int lo = i < c ? i : c, hi = i < c ? c : i;
        col[i] = (i == c ? 40.0 : 0.0) + 1.0 / (1.0 + lo + 2 * hi + (wid & 7));      // symmetric, diagonally dominant
    }
    double bj = 1.0 + c;
    double p_own, rp_own;
    bool bad = wl_factor<DP>(col, p_own, rp_own, tri, lane);
    const double sq = p_own * fast_rsqrt(p_own);
    const double bp = wl_forward<DP>(col, bj, rp_own, lane);
    const double yh = fma(bdf_normal(1234, *sweep, 1, 1, (uint64_t)wid, c), sq, bp);
    const double x = wl_backward<DP>(tri, yh, rp_own, lane);
    out[wid * 64 + lane] = bad ? -1.0 : x;
}

int main()
{
    double *out; uint32_t *sw;
    hipMalloc(&out, 64 * 8 * 70000); hipMalloc(&sw, 4); hipMemset(sw, 0, 4);
    for (int nw : {256, 1024, 2048, 3072, 4096, 6144, 12288, 24576}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_finish<32>, dim3((nw + 3) / 4), dim3(256), 0, 0, out, nw, sw);
        hipEventRecord(e0);
        for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_finish<32>, dim3((nw + 3) / 4), dim3(256), 0, 0, out, nw, sw);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("DP=32 waves=%6d  %.2f us per launch  (%.2f us per 1000 row pairs)\n", nw, ms * 100, ms * 100 * 1000 / nw);
    }
    double h[64];
    hipMemcpy(h, out, 512, hipMemcpyDeviceToHost);
    printf("x[0..3] = %.6f %.6f %.6f %.6f\n", h[0], h[1], h[2], h[3]);
    return 0;
}
