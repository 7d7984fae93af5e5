// The row kernel's finish phase by itself: every wave factors REPS synthetic 32 x 32 systems held in the accumulator layout
// (factor_all<32>, the forward solve riding along) and runs the backward solve (backward_all<32>), with 1 .. 8 waves
// resident per SIMD (LDS ballast).  w = 1 gives the length of one wave's dependent chain, large w the issue-bound rate:
// what a launch of N rows can reach at w waves per SIMD is max(chain, w x issue) x N / (1024 w).
//   hipcc --offload-arch=gfx950 -O3 -Ibayesiandatafusion.jl_amd/csrc [-DPROBE_DP=64] -o factor_probe tools/factor_probe.hip && ./factor_probe
#include "c_layout_chol.h"
#include <cstdio>
#include <cstdlib>
void bdf_set_error(const char *, ...) {}

#ifndef PROBE_MINWAVES
#define PROBE_MINWAVES (DP == 64 ? 2 : 8)
#endif
template <int DP>
__global__ __launch_bounds__(256, PROBE_MINWAVES) void k_factor(int reps, double *out, int with_backward)
{
    using GG = Geo<DP>;
    constexpr int DB = GG::DB, NB = GG::NB;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, h = lane >> 4;
    #ifdef PROBE_ALIAS     // timing experiment: the workgroup's four waves share ONE factor region (results are garbage) so that more
                       // waves per SIMD fit than the real kernel's LDS allows -- what would 3 or 4 waves per SIMD buy at D = 64?
    double *tri = lds;
#else
    double *tri = lds + wave * GG::WAVE_LDS;
#endif
    double res = 0.0;
    for (int rep = 0; rep < reps; rep++) {
        double A[NB * 4], bv[DB], ts[DB];
#pragma unroll
        for (int I = 0; I < DB; I++)
#pragma unroll
            for (int J = 0; J <= I; J++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * I + h + 4 * r, col = 16 * J + j;
                    A[GG::blk(I, J) * 4 + r] = (row == col ? 40.0 + rep : 0.0) + 1.0 / (1.0 + row + 2 * col + (blockIdx.x & 7));
                }
#pragma unroll
        for (int J = 0; J < DB; J++) { bv[J] = 1.0 + 16 * J + j; ts[J] = 0.0; }
#ifdef PROBE_BLOCKED
        factor_all_blocked<DP>(A, bv, ts, tri, j, h, DP, std::make_integer_sequence<int, DP - 1>{});
#else
        factor_all<DP>(A, bv, ts, tri, j, h, DP, std::make_integer_sequence<int, DP - 1>{});
#endif
        wave_sync();
        double yh = 0.0;
        if (with_backward) {
            const typename GG::ColRT cr = GG::col_rt(lane < DP ? lane : 0);
            const double dv = tri[cr.cbase + (lane & 3) * cr.nr4];
            const double rdv = fast_rcp(dv);
            yh = (lane < DP) ? ts[lane >> 4] : 0.0;
            unsigned colq[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                colq[q] = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(tri + cr.cbase + q * cr.nr4 - cr.q);
            backward_all<DP>(yh, rdv, colq, std::make_integer_sequence<int, DP / 16>{});
            yh *= rdv;
        } else {
            yh = ts[0] + tri[lane];
        }
        res += yh;
        wave_sync();
    }
    if (res == 12345.678) out[threadIdx.x] = res;
}

int main()
{
    double *out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#ifndef PROBE_DP
#define PROBE_DP 32
#endif
    constexpr int DP = PROBE_DP;
    const int reps = 64;
    for (int bw = 0; bw < 2; bw++)
        for (int w : {1, 2, 3, 4, 5, 6, 7, 8}) {
            // workgroups of 4 waves (one per SIMD); w workgroups resident per CU through the LDS each one asks for
            #ifdef PROBE_ALIAS
            const size_t need = 1 * Geo<DP>::WAVE_LDS * sizeof(double);
#else
            const size_t need = 4 * Geo<DP>::WAVE_LDS * sizeof(double);
#endif
            size_t lds = (size_t)(160 * 1024 / w) / 64 * 64;
            if (lds < need) { printf("w=%d does not fit\n", w); continue; }
            if (w == 8) lds = need;
            (void)hipFuncSetAttribute((const void *)k_factor<DP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            float best = 1e9f;
            for (int r = 0; r < 4; r++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k_factor<DP>, dim3(256 * w), dim3(256), lds, 0, reps, out, bw);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r && ms < best) best = ms;
            }
            const double per = best * 1e3 / reps;                 // us per system for one wave
            printf("%s  waves/SIMD %d: %8.1f us per launch, one system per wave in %6.2f us = %6.0f cycles at 2.4 GHz; per SIMD one system every %5.2f us;"
                   " 6726 rows on 1024 SIMDs: %5.1f us (D = %d)\n", bw ? "factor+backward" : "factor only    ", w, best * 1e3, per, per * 2400.0, per / w,
                   per / w * 6726.0 / 1024.0, DP);
        }
    return 0;
}
