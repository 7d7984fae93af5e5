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
#ifndef BDF_K1_WAVES32M
#define BDF_K1_WAVES32M 6         // ... its variant for two-mode relations (78 registers)
#endif
#ifndef BDF_K1_WAVES32
#define BDF_K1_WAVES32 5          // waves per SIMD the D <= 32 kernel is compiled for
#endif
#ifndef BDF_K1_WAVES32C
#define BDF_K1_WAVES32C 7         // ... its variant for one two-mode relation with coded values (70 registers)
#endif

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int DP>
struct Geo {
    static constexpr int DB = DP / 16;                 // 16-wide blocks per dimension
    static constexpr int NB = DB * (DB + 1) / 2;       // lower block-triangle
    static constexpr int PSZ = NB * 4 * 64 + DB * 16;  // doubles per partial slot
    static constexpr int WPB = (DP == 64) ? 2 : BDF_K1_WPB;           // waves per workgroup
    static constexpr int WAVES = (DP == 64) ? 2 : (DP == 32 ? BDF_K1_WAVES32 : 8);
    static constexpr int WAVES_MATRIX = (DP == 64) ? 2 : (DP == 32 ? BDF_K1_WAVES32M : 8);     // the two-mode-only variant
    static constexpr int WAVES_CODED = (DP == 64) ? 2 : (DP == 32 ? BDF_K1_WAVES32C : 8);      // one two-mode relation, coded values
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
    static constexpr int WAVE_LDS = TRI_D;
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

}  // namespace
