// dpp_probe.hip -- checks v_fmac_f64_dpp row_newbcast semantics on gfx950 and measures its issue rate against a plain
// v_fmac_f64 (both: 8 independent accumulators, 4096 rounds, one wave per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool DPP>
__global__ void k_rate(double *out, unsigned long long *cyc, int rounds)
{
    double a[8], s = 1.0 + threadIdx.x * 1e-9, m = 1e-9;
    for (int i = 0; i < 8; i++) a[i] = i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; r++) {
        if (DPP)
            asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %7, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(s), "v"(m));
        else
            asm volatile("v_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %2, %8, %9\n\tv_fmac_f64 %3, %8, %9\n\t"
                         "v_fmac_f64 %4, %8, %9\n\tv_fmac_f64 %5, %8, %9\n\tv_fmac_f64 %6, %8, %9\n\tv_fmac_f64 %7, %8, %9"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(s), "v"(m));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double t = 0;
    for (int i = 0; i < 8; i++) t += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_sem(double *out)
{
    double a = 0.0, s = (double)threadIdx.x, m = 1.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(s), "v"(m));
    out[threadIdx.x] = a;
}

__global__ void k_spin(unsigned long long ticks, unsigned long long *o)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) { }
    o[0] = __builtin_amdgcn_s_memtime() - t0;
    o[1] = __builtin_amdgcn_s_memrealtime() - r0;
}

__global__ void k_rcp(double *o)
{
    // accuracy of v_rcp_f64 / v_rsq_f64 alone and after one Newton step
    const double x = 0.37 + threadIdx.x * 1.618033988749;
    double y = __builtin_amdgcn_rcp(x);
    o[threadIdx.x * 4 + 0] = fabs(y * x - 1.0);
    double e = fma(-x, y, 1.0); y = fma(y, e, y);
    o[threadIdx.x * 4 + 1] = fabs(fma(y, x, -1.0));
    double r = __builtin_amdgcn_rsq(x);
    o[threadIdx.x * 4 + 2] = fabs(r * r * x - 1.0);
    double e2 = fma(-x * r, r, 1.0); r = fma(r * 0.5, e2, r);
    o[threadIdx.x * 4 + 3] = fabs(fma(r * r, x, -1.0));
}

int main()
{
    {
        double *o; hipMalloc(&o, 256 * 4 * 8);
        k_rcp<<<1, 256>>>(o);
        std::vector<double> h(1024); hipMemcpy(h.data(), o, 8192, hipMemcpyDeviceToHost);
        double m[4] = {0, 0, 0, 0};
        for (int i = 0; i < 256; i++) for (int q = 0; q < 4; q++) m[q] = h[i * 4 + q] > m[q] ? h[i * 4 + q] : m[q];
        printf("max relative error: v_rcp_f64 %.3g, +1 Newton %.3g; v_rsq_f64 %.3g, +1 Newton %.3g\n", m[0], m[1], m[2], m[3]);
    }
    {   // what does s_memtime count?  spin for 2^28 ticks and compare with the wall clock and s_memrealtime (100 MHz)
        unsigned long long *o; hipMalloc(&o, 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_spin<<<1, 64>>>(1000, o); hipDeviceSynchronize();
        hipEventRecord(e0); k_spin<<<1, 64>>>(1ull << 28, o); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
        printf("s_memtime: %llu ticks in %.3f ms -> %.1f MHz; s_memrealtime %llu ticks -> %.1f MHz\n", h[0], ms, h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
    }
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 1 << 16);
    k_sem<<<1, 64>>>(out);
    std::vector<double> h(64);
    hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) bad += h[l] != (double)((l & ~15) + 5);
    printf("row_newbcast:5 semantics %s (lane 0 -> %g, lane 20 -> %g, lane 63 -> %g)\n", bad ? "WRONG" : "ok", h[0], h[20], h[63]);
    const int rounds = 4096;
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int dpp = 0; dpp < 2; dpp++) {
            if (dpp) k_rate<true><<<1, 256 * waves>>>(out, cyc, rounds); else k_rate<false><<<1, 256 * waves>>>(out, cyc, rounds);
            unsigned long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s, %d wave(s)/SIMD: %.2f cycles per instruction per wave\n", dpp ? "v_fmac_f64_dpp" : "v_fmac_f64    ", waves,
                   (double)c / (rounds * 8.0));
        }
    }
    // chip-wide fp64 FMA rate: 256 CUs x {1,2,4,8} workgroups of 256 threads (= waves per SIMD), 8 independent chains each
    for (int wps = 1; wps <= 8; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = 256 * wps, rounds2 = 1 << 16;
        k_rate<false><<<blocks, 256>>>(out, cyc, 1024);
        hipDeviceSynchronize();
        hipEventRecord(e0); k_rate<false><<<blocks, 256>>>(out, cyc, rounds2); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 64 * 8.0 * rounds2 * 4.0 * blocks;
        printf("%d wave(s)/SIMD on every CU: %.1f TFLOP/s fp64 FMA (%.3f ms)\n", wps, flops / ms / 1e9, ms);
    }
    return bad;
}
