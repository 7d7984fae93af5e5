// Probe: gather rows with __builtin_amdgcn_global_load_lds (per-lane source address, lane-linear LDS destination).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__global__ void k_gather(const double *table, const int *idx, int D, double *out)
{
    __shared__ double buf[2][128];      // 2 slots x 1 KB
    const int lane = threadIdx.x;
    for (int s = 0; s < 2; s++) {
        const int row = idx[s * 4 + (lane >> 4)];
        const char *src = (const char *)(table + (size_t)row * D) + (lane & 15) * 16;
        __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)&buf[s][0], 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < 2; s++)
        for (int e = lane; e < 128; e += 64) out[s * 128 + e] = buf[s][e];
}

int main()
{
    const int D = 32, N = 100;
    std::vector<double> t(N * D);
    for (int i = 0; i < N * D; i++) t[i] = i;
    int hidx[8] = {5, 99, 0, 42, 7, 7, 63, 1};
    double *dt, *dout; int *didx;
    hipMalloc(&dt, t.size() * 8); hipMalloc(&dout, 256 * 8); hipMalloc(&didx, 32);
    hipMemcpy(dt, t.data(), t.size() * 8, hipMemcpyHostToDevice); hipMemcpy(didx, hidx, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_gather, dim3(1), dim3(64), 0, 0, dt, didx, D, dout);
    std::vector<double> o(256);
    hipMemcpy(o.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int s = 0; s < 2; s++) for (int r = 0; r < 4; r++) for (int e = 0; e < 32; e++)
        if (o[s * 128 + r * 32 + e] != (double)(hidx[s * 4 + r] * D + e)) bad++;
    printf("dma gather mismatches: %d of 256\n", bad);
    return 0;
}
