// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand/result lane maps (exact integer data) and issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double *A, const double *B, double *C)   // A 16x4 row-major, B 4x16 row-major, C 16x16 row-major
{
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

__global__ void k_rate(double *out, int iters, long long *cyc)
{
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void k_rate_fma(double *out, int iters, long long *cyc)
{
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    double c[16];
    for (int i = 0; i < 16; i++) c[i] = i;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) c[j] = fma(a, b, c[j]);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 16; i++) s += c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    std::vector<double> A(64), B(64), C(256), Ce(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i * 7 + k * 3;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 + k * 5 + j * 11 + (j * j) % 7;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) Ce[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC; long long *dcyc;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 1 << 22); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) if (C[i] != Ce[i]) bad++;
    printf("layout mismatches: %d of 256\n", bad);
    for (int waves = 1; waves <= 8; waves *= 2) {
        int iters = 20000; long long cyc;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        // one block per CU-ish: 256 blocks x (waves*4*64) threads -> `waves` waves per SIMD
        hipLaunchKernelGGL(k_rate, dim3(256), dim3(64 * 4 * waves > 1024 ? 1024 : 64 * 4 * waves), 0, 0, dC, 10, dcyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate, dim3(256 * (64 * 4 * waves > 1024 ? waves / 4 : 1)), dim3(64 * 4 * waves > 1024 ? 1024 : 64 * 4 * waves), 0, 0, dC, iters, dcyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
        double mf = 256.0 * 4 * waves * iters * 4;      // mfma instructions chip-wide
        printf("mfma f64: waves/SIMD=%d  %.1f clk64-cycles per MFMA (one wave)  chip: %.2f TFLOP/s  (%.3f ms)\n", waves,
               (double)cyc / (iters * 4.0), mf * 2048 / (ms * 1e-3) / 1e12, ms);
        hipLaunchKernelGGL(k_rate_fma, dim3(256), dim3(64 * 4 * waves > 1024 ? 1024 : 64 * 4 * waves), 0, 0, dC, 10, dcyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate_fma, dim3(256 * (64 * 4 * waves > 1024 ? waves / 4 : 1)), dim3(64 * 4 * waves > 1024 ? 1024 : 64 * 4 * waves), 0, 0, dC, iters, dcyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
        double nf = 256.0 * 4 * waves * iters * 16 * 64;   // lane-FMAs chip-wide
        printf("v_fma_f64: waves/SIMD=%d  %.1f clk64-cycles per FMA instr (one wave)  chip: %.2f TFLOP/s  (%.3f ms)\n", waves,
               (double)cyc / (iters * 16.0), nf * 2 / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
