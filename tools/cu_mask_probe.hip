// hipExtStreamCreateWithCUMask on gfx950: which CUs does a mask select, do a masked stream and its complement really run on
// disjoint CUs, and can a 1-workgroup kernel on a stream with one CU per XCD start while a chip-filling kernel runs?
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask_probe tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void where(unsigned *out, long long spin)
{
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);      // XCC_ID
        out[blockIdx.x] = (xcc & 0xf) << 16 | ((hw >> 13) & 0x7) << 12 | ((hw >> 12) & 1) << 8 | ((hw >> 8) & 0xf);   // xcc | se | sh | cu
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char **argv)
{
    int reserve = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> small(words, 0), big(words, 0);
    for (int i = 0; i < ncu; i++) (i < reserve ? small : big)[i / 32] |= 1u << (i % 32);
    hipStream_t s_small, s_big;
    printf("CUs %d, reserve %d: create small %d big %d\n", ncu, reserve, (int)hipExtStreamCreateWithCUMask(&s_small, words, small.data()),
           (int)hipExtStreamCreateWithCUMask(&s_big, words, big.data()));
    unsigned *o1, *o2;
    hipMalloc(&o1, 4096 * 4); hipMalloc(&o2, 65536 * 4);
    std::vector<unsigned> h1(4096), h2(65536);
    hipLaunchKernelGGL(where, dim3(64), dim3(64), 0, s_small, o1, 100LL);
    hipLaunchKernelGGL(where, dim3(8192), dim3(256), 0, s_big, o2, 100LL);
    hipDeviceSynchronize();
    hipMemcpy(h1.data(), o1, 64 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), o2, 8192 * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> a(h1.begin(), h1.begin() + 64), b(h2.begin(), h2.begin() + 8192);
    int common = 0;
    for (unsigned x : a) common += b.count(x);
    printf("small stream ran on %zu distinct CUs:", a.size());
    for (unsigned x : a) printf(" x%u.se%u.sh%u.cu%u", x >> 16, (x >> 12) & 7, (x >> 8) & 1, x & 0xf);
    printf("\nbig stream ran on %zu distinct CUs; common with small: %d\n", b.size(), common);
    // latency of a 1-workgroup kernel on the small stream while a chip-filling kernel (8 waves per SIMD, 2 ms) runs on the big one
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int masked = 0; masked < 2; masked++) {
        hipStream_t sb = masked ? s_big : 0, ss = masked ? s_small : nullptr;
        hipStream_t plain;
        hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
        if (!masked) ss = plain;
        hipLaunchKernelGGL(where, dim3(16384), dim3(256), 0, sb, o2, 20000LL);     // 200 us per wave, several generations
        hipEventRecord(e0, ss);
        hipLaunchKernelGGL(where, dim3(1), dim3(256), 0, ss, o1, 100LL);
        hipEventRecord(e1, ss);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: 1-workgroup kernel beside a chip-filling one took %.1f us\n", masked ? "masked streams" : "plain streams", ms * 1e3);
        hipDeviceSynchronize();
    }
    return 0;
}
