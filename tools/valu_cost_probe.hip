// valu_cost_probe.hip -- what one instruction of each kind costs a SIMD (gfx950): streams of 8 independent instructions per
// round, w = 1, 2, 4, 8 waves per SIMD on every CU, wall clock -> cycles per instruction per SIMD at 2.4 GHz.  The cost model
// behind the finish phase of the row kernel: which of the fp64 forms are full rate, what the cross-lane
// and scalar-read forms cost, and whether matrix and vector fp64 instructions of different waves overlap.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_cost_probe.hip -o tools/bin/valu_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

#define R8(S) S S S S S S S S
template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, int rounds)
{
    __shared__ double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i;
    __syncthreads();
    double a[8], s = 1.0 + threadIdx.x * 1e-9, m = 1e-9, t = 0.5;
    float f[8], fs = 1.0f + threadIdx.x * 1e-6f, fm = 1e-6f;
    d4 acc[4];
    unsigned u[8];
    for (int i = 0; i < 8; i++) { a[i] = i; f[i] = i; u[i] = i + threadIdx.x; }
    for (int i = 0; i < 4; i++) acc[i] = d4{0, 0, 0, 0};
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds + (threadIdx.x >> 4) * 64;
#define A8 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define F8 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
#define U8 "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
    for (int r = 0; r < rounds; r++) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                                    "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7" : F8 : "v"(fs), "v"(fm));
        if (KIND == 1) asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                                    "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7" : A8 : "v"(s), "v"(m));
        if (KIND == 2) asm volatile("v_fmac_f64 %0, %8, %9\n v_fmac_f64 %1, %8, %9\n v_fmac_f64 %2, %8, %9\n v_fmac_f64 %3, %8, %9\n"
                                    "v_fmac_f64 %4, %8, %9\n v_fmac_f64 %5, %8, %9\n v_fmac_f64 %6, %8, %9\n v_fmac_f64 %7, %8, %9" : A8 : "v"(s), "v"(m));
        if (KIND == 3) asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                                    "v_fmac_f64_dpp %2, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                                    "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                                    "v_fmac_f64_dpp %6, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf" : A8 : "v"(s), "v"(m));
        if (KIND == 4) asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                                    "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : A8 : "v"(s));
        if (KIND == 5) asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                                    "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : A8 : "v"(m));
        if (KIND == 6) asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n"
                                    "v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8" : A8 : "v"(s));
        if (KIND == 7) { unsigned s0, s1, s2, s3, s4, s5, s6, s7;
                         asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %10, 7\n v_readlane_b32 %3, %11, 9\n"
                                      "v_readlane_b32 %4, %8, 11\n v_readlane_b32 %5, %9, 13\n v_readlane_b32 %6, %10, 15\n v_readlane_b32 %7, %11, 17"
                                      : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
                         u[4] += s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7; }
        if (KIND == 8) asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                                    "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : A8);
        if (KIND == 9 || KIND == 14 || KIND == 15 || KIND == 16) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s, m, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s, m, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(s, m, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(s, m, acc[3], 0, 0, 0);
            if (KIND == 14) asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                                    "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7" : A8 : "v"(s), "v"(m));
            if (KIND == 15) asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                                    "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7" : F8 : "v"(fs), "v"(fm));
            if (KIND == 16) { double q0, q1, q2, q3;
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(addr) : "memory");
                t += q0 + q1; a[0] += q2 + q3; }
        }
        if (KIND == 10) { double q[8];
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                         "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)"
                         : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3]), "=v"(q[4]), "=v"(q[5]), "=v"(q[6]), "=v"(q[7]) : "v"(addr) : "memory");
            t += q[0]; addr ^= 8; }
        if (KIND == 11) { d4 q[4];   // 4 x ds_read_b128 = 8 doubles
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 p[4];
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=v"(p[0]), "=v"(p[1]), "=v"(p[2]), "=v"(p[3]) : "v"(addr) : "memory");
            t += p[0][0]; addr ^= 16; (void)q; }
        if (KIND == 12) asm volatile("v_mul_lo_u32 %0, %0, %0\n v_mul_lo_u32 %1, %1, %1\n v_mul_lo_u32 %2, %2, %2\n v_mul_lo_u32 %3, %3, %3\n"
                                     "v_mul_hi_u32 %4, %4, %4\n v_mul_hi_u32 %5, %5, %5\n v_mul_hi_u32 %6, %6, %6\n v_mul_hi_u32 %7, %7, %7" : U8);
        if (KIND == 13) asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %4\n"
                                     "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0" : U8);
        if (KIND == 17) asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                                    "v_fmac_f64_dpp %2, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                                    "s_nop 1\n v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                                    "v_fmac_f64_dpp %6, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf" : A8 : "v"(s), "v"(m));
        if (KIND == 18) { unsigned long long sv;     // exec-masked store pattern of the factorisation's owner lanes: 2 x (4 salu + 2 ds_write2_b64)
            asm volatile("s_mov_b64 %0, exec\n s_mov_b32 exec_lo, 0x10001\n s_mov_b32 exec_hi, 0x10001\n ds_write2_b64 %1, %2, %3 offset0:0 offset1:1\n ds_write2_b64 %1, %4, %5 offset0:2 offset1:3\n s_mov_b64 exec, %0\n"
                         "s_mov_b64 %0, exec\n s_mov_b32 exec_lo, 0x20002\n s_mov_b32 exec_hi, 0x20002\n ds_write2_b64 %1, %2, %3 offset0:4 offset1:5\n ds_write2_b64 %1, %4, %5 offset0:6 offset1:7\n s_mov_b64 exec, %0"
                         : "=&s"(sv) : "v"(addr), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory"); }
    }
    for (int i = 0; i < 8; i++) t += a[i] + f[i] + u[i];
    for (int i = 0; i < 4; i++) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678) out[threadIdx.x] = t;
}

template <int KIND>
void run(const char *name, double per_round, double *out)
{
    const int rounds = 20000;
    printf("%-44s", name);
    for (int w : {1, 2, 4, 8}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * w), dim3(256), 0, 0, out, 200);
        (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<KIND>, dim3(256 * w), dim3(256), 0, 0, out, rounds);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // per SIMD: w waves each ran rounds * per_round instructions
        printf("  w=%d %6.2f", w, best * 1e-3 * 2.4e9 / (rounds * per_round * w));
    }
    printf("   cycles per instruction per SIMD @2.4 GHz\n");
}

int main()
{
    double *out; (void)hipMalloc(&out, 1 << 20);
    run<0>("v_fma_f32", 8, out);
    run<1>("v_fma_f64", 8, out);
    run<2>("v_fmac_f64", 8, out);
    run<3>("v_fmac_f64_dpp row_newbcast", 8, out);
    run<17>("v_fmac_f64_dpp, s_nop 1 before each 4", 8, out);
    run<4>("v_mul_f64", 8, out);
    run<5>("v_add_f64", 8, out);
    run<6>("v_mov_b64", 8, out);
    run<7>("v_readlane_b32", 8, out);
    run<8>("v_rcp_f64", 8, out);
    run<12>("v_mul_lo/hi_u32", 8, out);
    run<13>("v_xor/v_add_u32", 8, out);
    run<9>("v_mfma_f64_16x16x4", 4, out);
    run<14>("(mfma_f64 + 2 v_fma_f64) per unit", 4, out);
    run<15>("(mfma_f64 + 2 v_fma_f32) per unit", 4, out);
    run<16>("(mfma_f64 + 1 ds_read_b64 + wait) per unit", 4, out);
    run<10>("ds_read_b64 x8 + wait (16-lane broadcast)", 8, out);
    run<11>("ds_read_b128 x4 + wait", 4, out);
    run<18>("owner-store group (4 salu + 2 ds_write2_b64)", 2, out);
    return 0;
}
