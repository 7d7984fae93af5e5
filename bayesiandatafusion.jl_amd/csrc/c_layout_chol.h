// c_layout_chol.h -- factorisation and triangular solves of a D x D (D <= 64) symmetric positive-definite matrix held
// in the MFMA accumulator (C) layout of one wavefront, shared by K1 (k_sample_rows.hip) and K6 (k_hyper.hip).
// Lane (j = l & 15, h = l >> 4), register r of block (I, J) holds element (16 I + h + 4 r, 16 J + j).  See the header of
// k_sample_rows.hip for the scheme (DPP row broadcasts folded into the fmas, owner lanes store each finished column to the
// packed factor in LDS under a compile-time EXEC mask, the forward solve rides along as one more matrix row).
#pragma once
#include "bdf_common.h"
#include "wave_linalg.h"
#include <utility>

#ifndef BDF_K1_WPB
#define BDF_K1_WPB 4
#endif
#ifndef BDF_K1_WPB64
#define BDF_K1_WPB64 1            // ... at D > 32: one (a wave's LDS comes free with it: with two, a finished wave's slot idled until its partner ended -- C4 42.7 -> 39.1 ms)
#endif
#ifndef BDF_K1_WAVES32M
#define BDF_K1_WAVES32M 6         // ... its variant for two-mode relations (78 registers)
#endif
#ifndef BDF_K1_WAVES32
#define BDF_K1_WAVES32 5          // waves per SIMD the D <= 32 kernel is compiled for
#endif
#ifndef BDF_K1_WAVES64
#define BDF_K1_WAVES64 2          // waves per SIMD the D > 32 kernel is compiled for (3: the 168-register build of round 6 -- the launch 2.4 % shorter, the sweep of C4 not: DESIGN 0 row 3)
#endif
#ifndef BDF_K1_WAVES32C
#define BDF_K1_WAVES32C 7         // ... its variant for one two-mode relation with coded values (70 registers)
#endif

namespace {

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant expression
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int DP>
struct Geo {
    static constexpr int DB = DP / 16;                 // 16-wide blocks per dimension
    static constexpr int NB = DB * (DB + 1) / 2;       // lower block-triangle
    static constexpr int PSZ = NB * 4 * 64 + DB * 16;  // doubles per partial slot
    static constexpr int WPB = (DP == 64) ? BDF_K1_WPB64 : BDF_K1_WPB;           // waves per workgroup
    static constexpr int WAVES = (DP == 64) ? BDF_K1_WAVES64 : (DP == 32 ? BDF_K1_WAVES32 : 8);
    static constexpr int WAVES_MATRIX = (DP == 64) ? BDF_K1_WAVES64 : (DP == 32 ? BDF_K1_WAVES32M : 8);     // the two-mode-only variant
    static constexpr int WAVES_CODED = (DP == 64) ? BDF_K1_WAVES64 : (DP == 32 ? BDF_K1_WAVES32C : 8);      // one two-mode relation, coded values
    __host__ __device__ static constexpr int blk(int I, int J) { return I * (I + 1) / 2 + J; }
    // packed factor in LDS: column k keeps rows col_first(k) = RG * (k / RG) .. DP-1, by row class:
    // entry i at col_base(k) + (i % 4) * col_rows(k) / 4 + (i - col_first(k)) / 4.  Columns are one double further apart
    // than they are long: an odd stride, so that the backward solve's per-lane reads (lane = column, same row) fall in
    // different LDS banks.  608 doubles at DP = 32 (4.9 KB per wave: seven waves per SIMD fit the 160 KB of a CU).
    static constexpr int RG = 4;
    __host__ __device__ static constexpr int col_first(int k) { return RG * (k / RG); }
    __host__ __device__ static constexpr int col_rows(int k) { return DP - col_first(k); }
    __host__ __device__ static constexpr int col_stride(int k) { return col_rows(k) + 1; }
    __host__ __device__ static constexpr int col_base(int k)
    {
        int s = 0;
        for (int q = 0; q < k / RG; q++) s += RG * col_stride(RG * q);
        return s + (k % RG) * col_stride(k);
    }
    static constexpr int TRI_D = (col_base(DP - 1) + col_stride(DP - 1) + 1) / 2 * 2;   // 176, 608, 2240 doubles
    // the same for a column known at run time (lane = column): base, rows per class, first row / 4
    struct ColRT { int cbase, nr4, q; };
    __device__ static inline ColRT col_rt(int c)
    {
        ColRT r;
        r.q = c / RG;
        r.nr4 = (DP - RG * r.q) / 4;
        r.cbase = RG * (r.q * (DP + 1) - RG * r.q * (r.q - 1) / 2) + (c % RG) * (DP - RG * r.q + 1);
        return r;
    }
    static constexpr int WAVE_LDS = TRI_D + 16;        // (+ the extra row's entries of one panel: factor_all_blocked)
};

// ---- DPP row-broadcast fma: d += (s of lane KJ of this lane's row of 16) * m.  Inline asm is opaque to the compiler's
// hazard recogniser (a VALU write of a DPP source needs 2 wait states before the DPP read), hence the leading s_nop. ----
template <int KJ>
__device__ __forceinline__ void fm4(double &d0, double &d1, double &d2, double &d3, double s0, double s1, double s2, double s3,
                           double m)
{
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                 : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(m), "n"(KJ));
}
template <int KJ>
__device__ __forceinline__ void fm4_self(double &d0, double &d1, double &d2, double &d3, double m)
{
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                 : "v"(m), "n"(KJ));
}
// the last 4 - R0 of a block's four registers (the diagonal block of the step's own block column: register r holds rows
// 4 r .. 4 r + 3 of the block, those at or above the pivot are finished -- nothing reads them again)
template <int KJ, int R0>
__device__ __forceinline__ void fm_self_from(double *t, double m)
{
    if constexpr (R0 <= 0) {
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                     : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : "v"(m), "n"(KJ));
    } else if constexpr (R0 == 1) {
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %0, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                     : "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : "v"(m), "n"(KJ));
    } else if constexpr (R0 == 2) {
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %0, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(t[2]), "+v"(t[3]) : "v"(m), "n"(KJ));
    } else {
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                     : "+v"(t[3]) : "v"(m), "n"(KJ));
    }
}
template <int KJ>
__device__ __forceinline__ void fm1(double &d, double s, double m)
{
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(d) : "v"(s), "v"(m), "n"(KJ));
}
template <int KJ>
__device__ __forceinline__ void fm1_self(double &d, double m)
{
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                 : "+v"(d) : "v"(m), "n"(KJ));
}

// ---- the factorisation (unscaled: the stored column k is Lt[i][k] = L[i][k] sqrt(d_k), Lt[k][k] = d_k) ------------------
// A[blk(I,J)*4 + r] = element (16 I + h + 4 r, 16 J + j); bv[J] = entry 16 J + j of the extra row b (same in every h).
// Software-pipelined: step k first updates the block column that holds column k+1, then -- while the rest of step k's
// updates issue -- column k+1's pivot is broadcast, its four owner lanes store it to the packed factor in LDS, and every
// lane reads back the entries of its own columns' rows (the multipliers of step k+1), so that neither the reciprocal
// nor the LDS round trip sits between two steps.
struct FactorLanes {            // per-lane LDS addressing, kept opaque so that it stays in registers
    int wr0, h8;                // owner lanes: byte address of class h at q = 0, and 8 h (one double less per class per q)
    int rd0, j3;                // multiplier reads: (j & 3) * DP / 4 + (j >> 2), and j & 3
};

// preparation of step k: pivot (wave-uniform), column k to LDS by its four owner lanes, the lane's multipliers
// nm[J] = -(Lt[16 J + j][k] / d_k); and the owner lanes keep entry k of the extra row (t_k, final now) in ts.
// The owner lanes (j == k % 16) are a compile-time lane pattern, so their part runs under a constant EXEC mask set by
// scalar moves: no vector compares, no copies (the compiler's own ds_write2 wanted the register-resident matrix copied).
// LDS operations of a wave execute in order, and the compiler's wait before it uses its own reads covers these writes.
template <int NV>
struct OwnerStore;
#define BDF_OWNER_STORE(NV, STORES, OPS, NARGS)                                                                        \
    template <>                                                                                                       \
    struct OwnerStore<NV> {                                                                                           \
        template <int MASK, int OFF>                                                                                  \
        __device__ static inline void run(unsigned addr, double v0, double v1, double v2, double v3)                  \
        {                                                                                                             \
            unsigned long long save;                                                                                  \
            asm volatile("s_mov_b64 %0, exec\n\t"                                                                     \
                         "s_mov_b32 exec_lo, %6\n\t"                                                                  \
                         "s_mov_b32 exec_hi, %6\n\t" STORES "s_mov_b64 exec, %0"                                      \
                         : "=&s"(save) : "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(MASK), "n"(OFF) : "memory"); \
        }                                                                                                             \
    };
// the last NV of the four registers of a block (rows below the column's first stored row are left out); OFF in doubles
// from addr.  Two doubles per LDS instruction: with four active lanes the LDS pipe is paid per instruction, not per byte.
BDF_OWNER_STORE(4, "ds_write2_b64 %1, %2, %3 offset0:%7 offset1:%7+1\n\tds_write2_b64 %1, %4, %5 offset0:%7+2 offset1:%7+3\n\t", , )
BDF_OWNER_STORE(3, "ds_write_b64 %1, %3 offset:(%7)*8\n\tds_write2_b64 %1, %4, %5 offset0:%7+1 offset1:%7+2\n\t", , )
BDF_OWNER_STORE(2, "ds_write2_b64 %1, %4, %5 offset0:%7 offset1:%7+1\n\t", , )
BDF_OWNER_STORE(1, "ds_write_b64 %1, %5 offset:(%7)*8\n\t", , )
#undef BDF_OWNER_STORE
template <int MASK>
__device__ __forceinline__ void owner_keep(double &dst, double src)
{
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b32 exec_lo, %3\n\t"
                 "s_mov_b32 exec_hi, %3\n\t"
                 "v_mov_b64 %0, %2\n\t"
                 "s_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(save) : "v"(src), "n"(MASK));
}

template <int DP, int k, int... Is>
__device__ __forceinline__ void owner_store_all(const double (&A)[Geo<DP>::NB * 4], unsigned addr, std::integer_sequence<int, Is...>)
{
    using GG = Geo<DP>;
    constexpr int K = k / 16, MASK = 0x00010001 << (k % 16), q = GG::col_first(k) / 4;
    constexpr int r0 = q - 4 * K;                       // first stored register of block (K, K): rows >= col_first(k)
    // block (K + Is, K), registers r >= (Is == 0 ? r0 : 0), to class-local positions 4 (K + Is) + r - q
    // (addr already points at the column: offsets stay within the 8-bit range of ds_write2_b64)
    (OwnerStore<(Is == 0 ? 4 - r0 : 4)>::template run<MASK, 4 * (K + Is) + (Is == 0 ? r0 : 0) - q>(
         addr, A[GG::blk(K + Is, K) * 4], A[GG::blk(K + Is, K) * 4 + 1], A[GG::blk(K + Is, K) * 4 + 2],
         A[GG::blk(K + Is, K) * 4 + 3]), ...);
}

template <int DP, int k, bool HAVE_RD = false>
__device__ __forceinline__ void prep(const double (&A)[Geo<DP>::NB * 4], const double (&bv)[Geo<DP>::DB], double (&ts)[Geo<DP>::DB],
                            double *tri, const FactorLanes &fl, double (&nm)[Geo<DP>::DB], double rd_ahead = 0.0)
{
    using GG = Geo<DP>;
    constexpr int DB = GG::DB, K = k / 16, kj = k % 16, kh = kj % 4, kr = kj / 4, cb = GG::col_base(k);
    constexpr int MASK = 0x00010001 << kj;                     // lanes with (lane & 15) == kj, per 32-lane half
    constexpr int q = GG::col_first(k) / 4;                   // rows per class: DP / 4 - q
    owner_store_all<DP, k>(A, (unsigned)(fl.wr0 - q * fl.h8 + cb * 8), std::make_integer_sequence<int, DB - K>{});
    owner_keep<MASK>(ts[K], bv[K]);
    wave_sync();
    double raw[DB];
    const int ri = fl.rd0 - q * fl.j3 + (cb - q);             // row 16 J + j of column k is at ri + 4 J
#pragma unroll
    for (int J = K; J < DB; J++) raw[J] = tri[ri + 4 * J];
    const double rd = HAVE_RD ? rd_ahead : fast_rcp(readlane_f64(A[GG::blk(K, K) * 4 + kr], kj + 16 * kh));
#pragma unroll
    for (int J = K; J < DB; J++) nm[J] = -(raw[J] * rd);
}

// step k: the updates of columns > k.  Block columns J > K first: they read column k (block column K) through the DPP
// broadcast, and the update of block column K rewrites it.  No masking of finished columns (<= k) in block column K:
// their registers are dead (a column is read for the last time at its own step).
template <int DP, int k>
__device__ __forceinline__ void factor_step(double (&A)[Geo<DP>::NB * 4], double (&bv)[Geo<DP>::DB], double (&ts)[Geo<DP>::DB],
                                   double *tri, const FactorLanes &fl, int j, double (&nm)[Geo<DP>::DB])
{
    using GG = Geo<DP>;
    constexpr int DB = GG::DB;
    constexpr int K = k / 16, kj = k % 16;
#ifdef BDF_CHOL_LOOKAHEAD
    // the next pivot, d_(k+1) = a_(k+1,k+1) + Lt_(k+1,k) nm_(k+1): the very fma the update below performs on that element,
    // done ahead on wave-uniform copies so that its reciprocal is ready when the step ends
    constexpr int k1 = k + 1, K1 = k1 / 16, j1 = k1 % 16, h1 = j1 % 4, r1 = j1 / 4;
    const double rd1 = fast_rcp(fma(readlane_f64(A[GG::blk(K1, K) * 4 + r1], kj + 16 * h1), readlane_f64(nm[K1], j1),
                                    readlane_f64(A[GG::blk(K1, K1) * 4 + r1], j1 + 16 * h1)));
#endif
#pragma unroll
    for (int J = DB - 1; J > K; J--) {
#pragma unroll
        for (int I = J; I < DB; I++) {
            double *t = &A[GG::blk(I, J) * 4];
            const double *s = &A[GG::blk(I, K) * 4];
            fm4<kj>(t[0], t[1], t[2], t[3], s[0], s[1], s[2], s[3], nm[J]);
        }
        fm1<kj>(bv[J], bv[K], nm[J]);
    }
    if constexpr (kj < 15) {                      // block column K still has unfinished columns
#pragma unroll
        for (int I = K; I < DB; I++) {
            double *t = &A[GG::blk(I, K) * 4];
            if (I == K) fm_self_from<kj, (kj + 1) / 4>(t, nm[K]);      // the diagonal block: only the registers with rows below the pivot
            else fm4_self<kj>(t[0], t[1], t[2], t[3], nm[K]);
        }
        fm1_self<kj>(bv[K], nm[K]);
    }
#ifdef BDF_CHOL_LOOKAHEAD
    prep<DP, k + 1, true>(A, bv, ts, tri, fl, nm, rd1);
#else
    prep<DP, k + 1>(A, bv, ts, tri, fl, nm);
#endif
}

template <int DP, int... Ks>
__device__ __forceinline__ void factor_all(double (&A)[Geo<DP>::NB * 4], double (&bv)[Geo<DP>::DB], double (&ts)[Geo<DP>::DB],
                                  double *tri, int j, int h, int D, std::integer_sequence<int, Ks...>)
{
    using GG = Geo<DP>;
    FactorLanes fl;
    fl.wr0 = (int)(unsigned)(size_t)(__attribute__((address_space(3))) double *)(tri + h * (DP / 4));   // LDS byte address
    fl.h8 = 8 * h;
    fl.rd0 = (j & 3) * (DP / 4) + (j >> 2);
    fl.j3 = j & 3;
    asm volatile("" : "+v"(fl.wr0), "+v"(fl.h8), "+v"(fl.rd0), "+v"(fl.j3));
    double nm[GG::DB];
    prep<DP, 0>(A, bv, ts, tri, fl, nm);
    // steps 0 .. D-2 (the last column has nothing to update; padded columns are skipped).  One wave-uniform exit per
    // step out of straight-line code (a skip-and-rejoin per step would make every step a merge point of the whole
    // register-resident matrix)
    (void)(... && ((Ks + 1 < D) && (factor_step<DP, Ks>(A, bv, ts, tri, fl, j, nm), true)));
}

// ======================================================================================================================
// Blocked variant: 16-column panels, trailing update on the matrix cores, no LDS round trip inside a step (used for DP = 64).
//
// Step k of the plain variant updates EVERY block column at and to the right of the pivot's (42 fp64 instructions per step
// at D = 64) and needs a transposed multiplier per block column: the whole of column k goes to LDS under a four-lane EXEC
// mask -- 13 cycles of the CU's LDS store path per ds_write2_b64 whatever the number of active lanes
// (tools/valu_cost_probe.hip), eight of them in front of every step of the first panel -- and comes back through a read:
// a dependent chain of ~600 cycles per step that the two waves a SIMD holds at D = 64 cannot hide.  Here:
//  * step k touches the pivot's own block column K only: the diagonal block, the panel below it, the extra row;
//  * its multipliers need no transposition through memory: the unfinished part of the diagonal block stays SYMMETRIC in the
//    registers (the updates are symmetric and whole registers are updated), so the multiplier of column 16 K + j -- entry
//    (16 K + j, k) -- is also entry (k, 16 K + j): register k / 4 of the block in the lanes of row k % 4, one ds_bpermute
//    away from every lane row.  The lanes of the finished columns take a zero multiplier, so finished columns keep their
//    final values in the registers and the whole panel is stored to the packed factor ONCE, when it is complete, with
//    full-wave stores;
//  * the block columns to the right get the panel's sixteen rank-1 updates at once when the panel is complete:
//        A(I,J) -= L(I,K) D_K^-1 L(J,K)'           I >= J > K         four v_mfma_f64_16x16x4_f64 per block
//    with the operands read back from the packed factor in the MFMA's operand layout -- lane (i = l & 15, kk = l >> 4)
//    takes row 16 I + i of column c = 16 K + 4 s + kk, and 1 / d_c from the column's stored pivot.  The extra row b (the
//    forward solve) follows the same way: its entries of the later block columns take sum_c L(.,c) t_c / d_c from the same
//    operands (4 fmas per block and panel + one reduction over the four lane rows).
// (The f64 MFMA is no cheaper per flop than full-lane vector fmas -- they share a pipe, and 64 cycles of it buy 2048 flops
// against 14 x 128 -- so this is not about moving work to the matrix cores: it shortens what every step waits for.
// tools/factor_probe.hip; at D = 32, seven waves per SIMD, the plain variant is already bound by the pipe and stays.)
// v = 0 in the lanes whose position in their row of 16 is in M16
template <int M16>
__device__ __forceinline__ void zero_lanes(double &v)
{
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b32 exec_lo, %2\n\t"
                 "s_mov_b32 exec_hi, %2\n\t"
                 "v_mov_b64 %0, 0\n\t"
                 "s_mov_b64 exec, %1"
                 : "+v"(v), "=&s"(save) : "n"(M16 | (M16 << 16)));
}

struct PanelLanes {             // per-lane constants of the blocked variant (opaque: they stay in registers)
    int kk, i3, i2;             // operand layout: lane row, (l & 15) & 3, (l & 15) >> 2
    int j4;                     // 4 (l & 15): ds_bpermute address of the lane's own position in lane row 0
};

template <int DP>
struct GeoB { static constexpr int T_OFF = Geo<DP>::TRI_D; };       // the extra row's entries of one panel sit behind the packed factor

// LOCAL = true (k_rows at DP = 64 since round 6: three waves per SIMD need <= 13 KB of LDS per wave where the whole packed factor
// is 17.9 KB): LDS holds ONE panel at a time -- panel K at the offsets of the packed layout minus col_base(16 K), every panel from
// offset 0 on -- because the only reader of the packed factor during the factorisation is the panel's own trailing update.  The
// finished columns stay in the registers (blocked variant: the lanes of finished columns take a zero multiplier), and the backward
// solve (backward_rows) puts them to LDS again one BLOCK ROW at a time, in the order it walks the rows.
template <int DP>
struct GeoL {
    using GG = Geo<DP>;
    static constexpr int PANEL0 = GG::col_base(16 < DP ? 16 : DP - 1);   // doubles of panel 0, the largest (944 at DP = 64)
    static constexpr int RS = DP + 1;                                    // row stride of the backward solve's block row (odd: lane = column reads, no bank conflicts)
    static constexpr int ROWS = 16 * RS;                                 // one block row: 16 rows x DP columns
    static constexpr int T_OFF = PANEL0;                                 // the extra row's entries of the panel being finished
    static constexpr int PIV = (ROWS > PANEL0 + 16 ? ROWS : PANEL0 + 16);   // the DP pivots, behind both
    static constexpr int WAVE_LDS = PIV + DP;                            // 1,104 doubles at DP = 64 (8.6 KB)
    __host__ __device__ static constexpr int base(int K) { return GG::col_base(16 * K); }
};

// the multiplier of block column K for step k (column k is final; the block column has unfinished columns)
template <int DP, int k>
__device__ __forceinline__ void prep_b(const double (&A)[Geo<DP>::NB * 4], const PanelLanes &pl, double &nmK)
{
    using GG = Geo<DP>;
    constexpr int K = k / 16, kj = k % 16, kh = kj % 4, kr = kj / 4;
    if constexpr (kj < 15) {
        const double row = A[GG::blk(K, K) * 4 + kr];           // lanes (j, kh): entry (k, 16 K + j) == entry (16 K + j, k)
        const int lo = __builtin_amdgcn_ds_bpermute(pl.j4 + 64 * kh, __double2loint(row));
        const int hi = __builtin_amdgcn_ds_bpermute(pl.j4 + 64 * kh, __double2hiint(row));
        const double rd = fast_rcp(readlane_f64(row, kj + 16 * kh));
        nmK = -(__hiloint2double(hi, lo) * rd);
        // the finished columns (lanes j <= kj) take no further updates
        zero_lanes<(1 << (kj + 1)) - 1>(nmK);
    }
}

// panel K to the packed factor: the diagonal block (rows from each column's first stored row), the blocks below it, the
// extra row's entries of the panel (lane j: t of column 16 K + j)
template <int DP, int K, bool LOCAL = false>
__device__ __forceinline__ void panel_store(const double (&A)[Geo<DP>::NB * 4], const double (&bv)[Geo<DP>::DB], double *tri, int j, int h)
{
    using GG = Geo<DP>;
    constexpr int DB = GG::DB;
    const typename GG::ColRT cr = GG::col_rt(16 * K + j);
    double *dst = tri + cr.cbase + h * cr.nr4 - cr.q - (LOCAL ? GeoL<DP>::base(K) : 0);   // row 16 I + h + 4 r of column 16 K + j at dst[4 I + r]
    const int r0 = j >> 2;                                      // the column stores rows 16 K + 4 (j >> 2) and on
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (r >= r0) dst[4 * K + r] = A[GG::blk(K, K) * 4 + r];
#pragma unroll
    for (int I = K + 1; I < DB; I++)
#pragma unroll
        for (int r = 0; r < 4; r++) dst[4 * I + r] = A[GG::blk(I, K) * 4 + r];
    if (h == 0) tri[(LOCAL ? GeoL<DP>::T_OFF : GeoB<DP>::T_OFF) + j] = bv[K];
}

// panel K is complete: to the packed factor, then the trailing update of the block columns J > K and of the extra row's
// entries there
template <int DP, int K, bool LOCAL = false>
__device__ __forceinline__ void panel_end(double (&A)[Geo<DP>::NB * 4], double (&bv)[Geo<DP>::DB], double *tri, int j, int h,
                                 const PanelLanes &pl)
{
    using GG = Geo<DP>;
    constexpr int DB = GG::DB;
    static_assert(K + 1 < DB, "the last panel has nothing to its right");
    panel_store<DP, K, LOCAL>(A, bv, tri, j, h);
    wave_sync();
    // operands: lane (i, kk), k-step s: row 16 I + i of column c = 16 K + 4 s + kk, at
    //     col_base(c) + (i & 3) R / 4 + 4 (I - K) - s + (i >> 2),  R = DP - 16 K - 4 s rows stored, col_base(c) = col_base(c - kk) + kk (R + 1);
    // the column's pivot d_c: the first entry of its row class kk, at col_base(c) + kk R / 4
    double pb[DB];
#pragma unroll
    for (int I = K + 1; I < DB; I++) pb[I] = 0.0;
    asm volatile("s_nop 1" ::: "memory");        // (the accumulators were last written by inline-asm VALU instructions)
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int R = DP - 16 * K - 4 * s;
        const int base = GG::col_base(16 * K + 4 * s) - (LOCAL ? GeoL<DP>::base(K) : 0);
        const double *src = tri + (base - s) + pl.kk * (R + 1) + pl.i3 * (R / 4) + pl.i2;
        double a[DB], as[DB];
#pragma unroll
        for (int I = K + 1; I < DB; I++) a[I] = src[4 * (I - K)];
        const double nr = -fast_rcp(tri[base + pl.kk * (R + 1 + R / 4)]);
        const double tc = tri[(LOCAL ? GeoL<DP>::T_OFF : GeoB<DP>::T_OFF) + 4 * s + pl.kk];
#pragma unroll
        for (int I = K + 1; I < DB; I++) as[I] = a[I] * nr;
#pragma unroll
        for (int J = K + 1; J < DB; J++)
#pragma unroll
            for (int I = J; I < DB; I++) {
                d4 c = d4{A[GG::blk(I, J) * 4], A[GG::blk(I, J) * 4 + 1], A[GG::blk(I, J) * 4 + 2], A[GG::blk(I, J) * 4 + 3]};
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(as[I], a[J], c, 0, 0, 0);
                A[GG::blk(I, J) * 4] = c[0]; A[GG::blk(I, J) * 4 + 1] = c[1]; A[GG::blk(I, J) * 4 + 2] = c[2]; A[GG::blk(I, J) * 4 + 3] = c[3];
            }
#pragma unroll
        for (int I = K + 1; I < DB; I++) pb[I] = fma(as[I], tc, pb[I]);
    }
#pragma unroll
    for (int I = K + 1; I < DB; I++) {
        double v = pb[I];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        bv[I] += v;
    }
    // the matrix instructions' results are read next by inline-asm VALU instructions, which the compiler's hazard
    // recogniser cannot see into: wait out the last one's passes here (16 passes + 2)
#pragma unroll
    for (int b = 0; b < GG::NB * 4; b++) asm volatile("" : "+v"(A[b]));
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
}

template <int DP, int k, bool LOCAL = false>
__device__ __forceinline__ void factor_step_b(double (&A)[Geo<DP>::NB * 4], double (&bv)[Geo<DP>::DB], double *tri, const PanelLanes &pl,
                                     int j, int h, double &nmK)
{
    using GG = Geo<DP>;
    constexpr int DB = GG::DB, K = k / 16, kj = k % 16;
    if constexpr (kj < 15) {
        // the diagonal block first: the next step's multipliers and pivot come from it
        fm_self_from<kj, (kj + 1) / 4>(&A[GG::blk(K, K) * 4], nmK);
        double nm_next = 0.0;
        prep_b<DP, k + 1>(A, pl, nm_next);
        fm1_self<kj>(bv[K], nmK);
#pragma unroll
        for (int I = K + 1; I < DB; I++) {
            double *t = &A[GG::blk(I, K) * 4];
            fm4_self<kj>(t[0], t[1], t[2], t[3], nmK);
        }
        nmK = nm_next;
    } else {
        if constexpr (K + 1 < DB) panel_end<DP, K, LOCAL>(A, bv, tri, j, h, pl);
        prep_b<DP, k + 1>(A, pl, nmK);
    }
}

// (tri: Geo<DP>::WAVE_LDS doubles)
template <int DP, int... Ks>
__device__ __forceinline__ void factor_all_blocked(double (&A)[Geo<DP>::NB * 4], double (&bv)[Geo<DP>::DB], double (&ts)[Geo<DP>::DB],
                                          double *tri, int j, int h, int D, std::integer_sequence<int, Ks...>)
{
    using GG = Geo<DP>;
    PanelLanes pl;
    pl.kk = h; pl.i3 = j & 3; pl.i2 = j >> 2; pl.j4 = 4 * j;
    asm volatile("" : "+v"(pl.kk), "+v"(pl.i3), "+v"(pl.i2), "+v"(pl.j4));
    double nmK = 0.0;
    prep_b<DP, 0>(A, pl, nmK);
    (void)(... && ((Ks + 1 < D) && (factor_step_b<DP, Ks>(A, bv, tri, pl, j, h, nmK), true)));
    // the panel the factorisation ended in (the earlier ones were stored when they were complete)
    const int Kf = (D - 1) >> 4;
    if constexpr (GG::DB >= 1) { if (Kf == 0) panel_store<DP, 0>(A, bv, tri, j, h); }
    if constexpr (GG::DB >= 2) { if (Kf == 1) panel_store<DP, 1>(A, bv, tri, j, h); }
    if constexpr (GG::DB >= 3) { if (Kf == 2) panel_store<DP, 2>(A, bv, tri, j, h); }
    if constexpr (GG::DB >= 4) { if (Kf == 3) panel_store<DP, 3>(A, bv, tri, j, h); }
    // the extra row's entries stopped changing when their columns finished: they are the forward solve
#pragma unroll
    for (int J = 0; J < GG::DB; J++) ts[J] = bv[J];
}

// LOCAL variant (GeoL): one panel in LDS at a time; at the end the factor is in the registers (A) and the pivots in tri[GeoL::PIV + c]
template <int DP, int... Ks>
__device__ __forceinline__ void factor_all_blocked_local(double (&A)[Geo<DP>::NB * 4], double (&bv)[Geo<DP>::DB], double (&ts)[Geo<DP>::DB],
                                                double *tri, int j, int h, int D, std::integer_sequence<int, Ks...>)
{
    using GG = Geo<DP>;
    PanelLanes pl;
    pl.kk = h; pl.i3 = j & 3; pl.i2 = j >> 2; pl.j4 = 4 * j;
    asm volatile("" : "+v"(pl.kk), "+v"(pl.i3), "+v"(pl.i2), "+v"(pl.j4));
    double nmK = 0.0;
    prep_b<DP, 0>(A, pl, nmK);
    (void)(... && ((Ks + 1 < D) && (factor_step_b<DP, Ks, true>(A, bv, tri, pl, j, h, nmK), true)));
#pragma unroll
    for (int J = 0; J < GG::DB; J++) ts[J] = bv[J];
    // the pivots: element (c, c), c = 16 K + j, is register j >> 2 of block (K, K) in the lane of row class h == j & 3
    const int r = j >> 2;
    static_for<GG::DB>([&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        constexpr int b4 = GG::blk(K, K) * 4;
        const double d01 = r == 0 ? A[b4] : A[b4 + 1], d23 = r == 2 ? A[b4 + 2] : A[b4 + 3];
        const double d = r < 2 ? d01 : d23;
        if (h == (j & 3)) tri[GeoL<DP>::PIV + 16 * K + j] = d;
    });
}

// ---- backward solve Lt' x = yh with lane = column: lane c < i subtracts Lt[i][c] x_i, read from the packed factor ----
// The entries a lane needs (row i of its column, i = D-1 ... 1) do not depend on the solve, so they are read from LDS a
// batch of rows ahead, under a compile-time EXEC mask (lanes c < i; the others have no such entry), and the dependent
// chain of a step is only: multiply by the pivot's reciprocal, read lane i, one masked fma.  (A load + branch per step,
// as the plain loop compiles, leaves an LDS round trip in every link of the chain: 190 cycles per step against ~40.)
template <int i>
struct BwMask {
    static constexpr unsigned long long M = (i >= 64) ? ~0ull : ((1ull << i) - 1ull);
    static constexpr unsigned LO = (unsigned)(M & 0xffffffffull), HI = (unsigned)(M >> 32);
};

template <int i>
__device__ __forceinline__ void backward_load(double &L, unsigned addr)
{
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b32 exec_lo, %3\n\t"
                 "s_mov_b32 exec_hi, %4\n\t"
                 "ds_read_b64 %0, %2 offset:%5\n\t"
                 "s_mov_b64 exec, %1"
                 : "=v"(L), "=&s"(save) : "v"(addr), "n"(BwMask<i>::LO), "n"(BwMask<i>::HI), "n"((i >> 2) * 8) : "memory");
}

template <int i>
__device__ __forceinline__ void backward_fma(double &yh, double L, double xi)
{
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b32 exec_lo, %4\n\t"
                 "s_mov_b32 exec_hi, %5\n\t"
                 "v_fma_f64 %0, -%2, %3, %0\n\t"
                 "s_mov_b64 exec, %1"
                 : "+v"(yh), "=&s"(save) : "v"(L), "s"(xi), "n"(BwMask<i>::LO), "n"(BwMask<i>::HI));
}

// rows I0, I0 - 1, ..., I0 - N + 1 (those >= 1): their entries loaded first, then the steps.  No test against D: the rows
// of the padding (D <= i < DP) have x_i = 0 and entries 0 -- the factorisation never stores them, so the caller zeroes
// the packed factor once when D < DP (zero_packed_factor) -- and a wave-uniform skip per step would make every step a merge
// point of all the batch's registers.
template <int DP, int I0, int N, int... Ns>
__device__ __forceinline__ void backward_batch(double &yh, double rdv, const unsigned (&colq)[4], std::integer_sequence<int, Ns...>)
{
    double L[N];
    (((I0 - Ns >= 1) ? backward_load<(I0 - Ns >= 1 ? I0 - Ns : 1)>(L[Ns], colq[(I0 - Ns) & 3]) : (void)0), ...);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    (((I0 - Ns >= 1)
          ? backward_fma<(I0 - Ns >= 1 ? I0 - Ns : 1)>(yh, L[Ns], readlane_f64(yh * rdv, (I0 - Ns >= 1 ? I0 - Ns : 1)))
          : (void)0), ...);
}

// the packed factor of a wave, zeroed (needed once per wave when D < DP, see backward_batch)
template <int DP>
__device__ __forceinline__ void zero_packed_factor(double *tri, int lane)
{
    for (int e = lane; e < Geo<DP>::TRI_D; e += 64) tri[e] = 0.0;
    wave_sync();
}

template <int DP, int... Bs>
__device__ __forceinline__ void backward_all(double &yh, double rdv, const unsigned (&colq)[4], std::integer_sequence<int, Bs...>)
{
    constexpr int N = 16;                             // rows per batch (32 registers of entries in flight)
    (backward_batch<DP, DP - 1 - N * Bs, N>(yh, rdv, colq, std::make_integer_sequence<int, N>{}), ...);
}

// ---- backward solve for the LOCAL variant: Lt' x = yh, lane = column, the factor coming from the REGISTERS one block row at a time:
// block row I (rows 16 I .. 16 I + 15 of every column up to the diagonal block) goes to LDS row-major with stride DP + 1, then the
// sixteen steps of those rows run as in backward_batch -- all sixteen rows' entries of the lane's column loaded first, then per step
// one multiply, one v_readlane, one masked fma.  Same operations in the same order as backward_all: the same bits.
template <int i, int RS>
__device__ __forceinline__ void backward_load_row(double &L, unsigned addr)
{
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b32 exec_lo, %3\n\t"
                 "s_mov_b32 exec_hi, %4\n\t"
                 "ds_read_b64 %0, %2 offset:%5\n\t"
                 "s_mov_b64 exec, %1"
                 : "=v"(L), "=&s"(save) : "v"(addr), "n"(BwMask<i>::LO), "n"(BwMask<i>::HI), "n"((i & 15) * RS * 8) : "memory");
}

template <int DP, int I, int... Ns>
__device__ __forceinline__ void backward_block_row(const double (&A)[Geo<DP>::NB * 4], double &yh, double rdv, double *rows, unsigned addr,
                                          int j, int h, std::integer_sequence<int, Ns...>)
{
    using GG = Geo<DP>;
    constexpr int RS = GeoL<DP>::RS;
    // (LDS operations of a wave execute in order: these writes cannot pass the previous block row's reads)
#pragma unroll
    for (int J = 0; J <= I; J++)
#pragma unroll
        for (int r = 0; r < 4; r++) rows[(h + 4 * r) * RS + 16 * J + j] = A[GG::blk(I, J) * 4 + r];
    wave_sync();
    double L[16];
    (((16 * I + 15 - Ns >= 1) ? backward_load_row<(16 * I + 15 - Ns >= 1 ? 16 * I + 15 - Ns : 1), RS>(L[Ns], addr) : (void)0), ...);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    (((16 * I + 15 - Ns >= 1)
          ? backward_fma<(16 * I + 15 - Ns >= 1 ? 16 * I + 15 - Ns : 1)>(yh, L[Ns], readlane_f64(yh * rdv, (16 * I + 15 - Ns >= 1 ? 16 * I + 15 - Ns : 1)))
          : (void)0), ...);
}

template <int DP>
__device__ __forceinline__ void backward_rows(const double (&A)[Geo<DP>::NB * 4], double &yh, double rdv, double *tri, int lane)
{
    const int j = lane & 15, h = lane >> 4;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(tri + lane);
    if constexpr (Geo<DP>::DB >= 4) backward_block_row<DP, 3>(A, yh, rdv, tri, addr, j, h, std::make_integer_sequence<int, 16>{});
    if constexpr (Geo<DP>::DB >= 3) backward_block_row<DP, 2>(A, yh, rdv, tri, addr, j, h, std::make_integer_sequence<int, 16>{});
    if constexpr (Geo<DP>::DB >= 2) backward_block_row<DP, 1>(A, yh, rdv, tri, addr, j, h, std::make_integer_sequence<int, 16>{});
    backward_block_row<DP, 0>(A, yh, rdv, tri, addr, j, h, std::make_integer_sequence<int, 16>{});
}


}  // namespace
