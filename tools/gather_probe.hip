// What rate can a row kernel's gather reach?  A table of R factor rows (D = 32 doubles = 256 B each, L2-resident: R = 3952
// or 6040 as on MovieLens) and a random index list; one wave walks `per_wave` indices 8 at a time in the row kernel's own
// access shape and adds what it loads (no MFMA, next to no FP64 work).  Variants:
//   VEC = 1: a lane loads 8 B -- element (l & 15) + 16 I of observation (l >> 4) (+ 4 for the second k-step): 4 loads per
//            lane and trip, each instruction 4 rows x 128 B (the row kernel's gather)
//   VEC = 2: a lane loads 16 B -- elements 2 (l & 15), 2 (l & 15) + 1 of observation (l >> 4): 2 loads per trip, each
//            instruction 4 whole rows (what an interleaved row layout would allow)
//   DEPTH  : trips in flight per wave
// and the number of resident waves per SIMD (dynamic LDS as ballast).  Prints GB/s of gathered rows.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe tools/gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int VEC, int DEPTH>
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ table, const int *__restrict__ idx, int per_wave, double *out)
{
    extern __shared__ double ballast[];
    const int lane = threadIdx.x & 63, j = lane & 15, h = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int *my = idx + wave * per_wave;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    const int trips = per_wave / 8;
    int id[DEPTH][2];
    double w[DEPTH][4];
    auto load_idx = [&](int t, int s) {
        const int tt = t < trips ? t : trips - 1;
        id[s][0] = my[tt * 8 + h];
        id[s][1] = my[tt * 8 + 4 + h];
    };
    auto load_rows = [&](int s) {
        if (VEC == 1) {
            const double *r0 = table + (int64_t)id[s][0] * 32, *r1 = table + (int64_t)id[s][1] * 32;
            w[s][0] = r0[j]; w[s][1] = r0[16 + j]; w[s][2] = r1[j]; w[s][3] = r1[16 + j];
        } else {
            const d2 a = *(const d2 *)(table + (int64_t)id[s][0] * 32 + 2 * j), b = *(const d2 *)(table + (int64_t)id[s][1] * 32 + 2 * j);
            w[s][0] = a[0]; w[s][1] = a[1]; w[s][2] = b[0]; w[s][3] = b[1];
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH; s++) load_idx(s, s);
#pragma unroll
    for (int s = 0; s < DEPTH - 1; s++) load_rows(s);
    for (int t0 = 0; t0 < trips; t0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; s++) {
            // slot s holds trip t0 + s: its rows were requested DEPTH - 1 trips ago; request the newest trip's rows, then the
            // indices DEPTH trips ahead
            load_rows((s + DEPTH - 1) % DEPTH);
            acc0 += w[s][0]; acc1 += w[s][1]; acc2 += w[s][2]; acc3 += w[s][3];
            load_idx(t0 + s + DEPTH, s);
        }
    }
    const double r = (acc0 + acc1) + (acc2 + acc3);
    if (r == 12345.678 || ballast[0] == 1e300) out[threadIdx.x] = r;
}

template <int VEC, int DEPTH>
static void run(const char *name, const double *table, const int *idx, int64_t n_obs, int per_wave, int waves_per_simd, double *out)
{
    const int64_t waves = n_obs / per_wave;
    const unsigned grid = (unsigned)(waves / 4);
    // LDS ballast: 160 KB per CU, workgroups of 4 waves (one per SIMD): waves_per_simd workgroups resident
    const size_t lds = waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / waves_per_simd - 512) / 16 * 16;
    hipFuncSetAttribute((const void *)k_gather<VEC, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_gather<VEC, DEPTH>), dim3(grid), dim3(256), lds, 0, table, idx, per_wave, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-18s waves/SIMD %d per_wave %4d: %7.1f us  %6.2f TB/s\n", name, waves_per_simd, per_wave, best * 1e3,
           (double)waves * per_wave * 256.0 / (best * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char **argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 3952;
    const int64_t n_obs = argc > 2 ? atoll(argv[2]) : 2 * 1024 * 1024;     // a multiple of 4 x per_wave
    std::vector<double> t((size_t)R * 32, 1.0);
    std::vector<int> ix(n_obs);
    uint64_t s = 88172645463325252ull;
    for (auto &v : ix) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % (uint64_t)R); }
    double *table, *out;
    int *idx;
    hipMalloc(&table, t.size() * 8); hipMalloc(&idx, ix.size() * 4); hipMalloc(&out, 4096);
    hipMemcpy(table, t.data(), t.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(idx, ix.data(), ix.size() * 4, hipMemcpyHostToDevice);
    printf("table %d rows x 256 B = %.2f MB, %lld gathered rows = %.0f MB per launch\n", R, R * 256.0 / 1e6, (long long)n_obs, n_obs * 256.0 / 1e6);
    for (int per_wave : {64, 128, 1024}) {
        for (int wps : {2, 4, 6, 8}) {
            run<1, 1>("8B  depth 1", table, idx, n_obs, per_wave, wps, out);
            run<1, 2>("8B  depth 2", table, idx, n_obs, per_wave, wps, out);
            run<1, 4>("8B  depth 4", table, idx, n_obs, per_wave, wps, out);
            run<1, 8>("8B  depth 8", table, idx, n_obs, per_wave, wps, out);
            run<2, 1>("16B depth 1", table, idx, n_obs, per_wave, wps, out);
            run<2, 2>("16B depth 2", table, idx, n_obs, per_wave, wps, out);
            run<2, 4>("16B depth 4", table, idx, n_obs, per_wave, wps, out);
            run<2, 8>("16B depth 8", table, idx, n_obs, per_wave, wps, out);
        }
    }
    return 0;
}
