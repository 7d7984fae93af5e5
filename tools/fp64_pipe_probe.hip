// Do v_mfma_f64_16x16x4_f64 and v_fma_f64 share an execution pipe on gfx950?  (MI355X: 78.6 TFLOP/s fp64 for BOTH the
// vector and the matrix path.)  One workgroup of 8 waves per CU = two waves per SIMD; mode 0: both run dependent-free
// v_fma_f64 streams, mode 1: both run MFMA streams, mode 2: one of each; modes 3 / 4: ONE wave per SIMD, fma / MFMA alone.  If the pipes were separate, mode 2 would take
// max(t_fma, t_mfma) per wave pair; if the MFMA executes on the vector unit's fp64 lanes, mode 2 takes the sum.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_pipe_probe tools/fp64_pipe_probe.hip && ./fp64_pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mode, int n, double *out)
{
    const int wave = threadIdx.x >> 6;
    if (mode >= 3 && wave >= 4) return;                     // one wave per SIMD
    const bool mfma = mode == 1 || mode == 4 || (mode == 2 && wave >= 4);
    const double x = 1.0 + 1e-9 * threadIdx.x, y = 1.0 - 1e-9 * threadIdx.x;
    double r = 0.0;
    if (mfma) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < n; i += 4) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        double c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        // 16 fmas per MFMA-equivalent (a 16x16x4 MFMA is 2048 flops = 16 wave-wide fmas): the same flops per `n`
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) c[u & 7] = fma(c[u & 7], x, y);
        }
        for (int u = 0; u < 8; u++) r += c[u];
    }
    if (r == 12345.678) out[threadIdx.x] = r;
}

int main()
{
    double *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 40000;
    const char *names[] = {"fma+fma", "mfma+mfma", "fma+mfma", "fma alone", "mfma alone"};
    for (int mode = 0; mode < 5; mode++) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, n, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // flops: waves x n x 2048 per CU x 256 CUs
        printf("mode %d (%s): %.3f ms, %.1f TFLOP/s chip-wide\n", mode, names[mode], ms, 256.0 * (mode >= 3 ? 4 : 8) * n * 2048.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
