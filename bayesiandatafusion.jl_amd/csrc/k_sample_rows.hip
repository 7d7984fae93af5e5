// k_sample_rows.hip -- K1: the latent-row sampler.
//
// Replaces sample_user_basic (src/sampling.jl:200-212 matrix, :215-234 tensor) and sample_user2
// (src/sampling.jl:266-289, sum over the entity's relations) of the reference, for every row of an
// entity at once (sample_latent_all2! :149-172, sample_user2_all! :251-264).
//
// Per row i:
//   S   = sum over the row's observations of w w',  w = Hadamard product of the other modes' factor rows
//   P_i = Lambda + sum_r alpha_r S_r          b_i = Lambda mu_i + sum_r alpha_r sum w (y - base)
//   x_i = chol(inv(P_i))' z + inv(P_i) b_i    (the reference's map from z to the sample)
//
// The reference forms inv(P_i) by LU and then a Cholesky factor of the covariance.  Here P_i is factored
// once as P_i = U U' with U UPPER triangular (a Cholesky factorisation run from the last index to the
// first).  Then inv(P_i) = U^-T U^-1 with U^-T lower triangular and positive diagonal, so by uniqueness
// of the Cholesky factor chol(inv(P_i))' == U^-T, and  x_i = U^-T (U^-1 b_i + z):  one factorisation and
// two triangular solves give exactly the reference's function of z (to fp64 rounding).
// All of it runs in index-reversed coordinates (e -> D-1-e), where U U' becomes an ordinary lower
// Cholesky L L' and the two solves become forward then backward substitution.
//
// Work decomposition (ragged rows: MovieLens rows have 0..1668 observations):
//   * a row's observations are cut into ITEMS of at most T observations; one wavefront accumulates one item.
//   * a row with a single item is DIRECT: the wave that accumulated it also factors, solves and draws.
//   * a row with several items (long rows, or several relations) is SPLIT: its items write partial (S, b) to a
//     scratch slab, and a second launch (k_rows_finish) adds a row's partials in slot order and finishes it.
//   so no wave ever owns more than T observations and results do not depend on scheduling.
//
// Accumulation: the rank-4 update S += W W' (W = D x 4 gathered factor rows) is one v_mfma_f64_16x16x4_f64 per
// 16x16 block of the lower block-triangle.  The MFMA A/B operand of lane l is element (l & 15) of observation
// (l >> 4): exactly what a coalesced 128-byte-per-16-lanes gather of the factor row delivers, so operands go from
// global memory to the matrix pipe with no LDS staging and no cross-lane traffic (measured on MI355X: 64 cycles
// per MFMA, 77 TFLOP/s chip-wide against 64 TFLOP/s for v_fma_f64 which would also need every operand broadcast).
//
// Finishing: G = 64/DP rows at a time per wave (lane group = row, lane in group = column of P~ held in registers).
// Cholesky keeps the matrix fully symmetric so row k of a lane's column doubles as A[c][k]; step k broadcasts row k
// through LDS once (ds_write_b64 + broadcast ds_reads) and updates the trailing rows with one fma per element.
#include "bdf_common.h"
#include "wave_linalg.h"
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

#ifndef BDF_K1_WAVES
#define BDF_K1_WAVES 2
#endif

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Item {             // one wave's accumulation work
    int32_t row;          // entity row (-1: padding)
    int32_t term;
    int64_t q_begin;      // first observation (index into the term's CSR arrays)
    int32_t count;        // observations in this item
    int32_t slot;         // partial slot, or -1 for a direct row
};

struct SplitRow {
    int32_t row;
    int32_t slot_begin, n_slots;
    int32_t _pad;
};

struct PlanDev {
    const Item *direct;   int32_t n_direct;      // padded to a multiple of G
    const Item *split;    int32_t n_split;
    const SplitRow *rows; int32_t n_split_rows;  // padded to a multiple of G
    double *partials;                            // n_split * PSZ doubles
};

template <int DP>
struct Geo {
    static constexpr int G = 64 / DP;                  // rows finished together by one wave
    static constexpr int DB = DP / 16;                 // 16-wide blocks per dimension
    static constexpr int NB = DB * (DB + 1) / 2;       // lower block-triangle
    static constexpr int LD = DP + 1;                  // padded leading dimension of the LDS images
    static constexpr int PSZ = NB * 4 * 64 + DB * 16;  // doubles per partial slot
    static constexpr int WPB = (DP == 64) ? 1 : 4;     // waves per workgroup (static LDS must stay under 64 KB)
    // LDS-DMA gather ring: a slot holds the 4 gathered factor rows of one MFMA k-step
    static constexpr int LPR = (DP == 64) ? 32 : 16;   // lanes (16 bytes each) per gathered row
    static constexpr int ROWB = LPR * 16;              // bytes between rows in a slot
    static constexpr int IPK = 4 * LPR / 64;           // DMA instructions per k-step and other mode
    static constexpr int SLOTB = 4 * ROWB;             // bytes per slot
    static constexpr int RING_SLOTS = 8192 / SLOTB;    // 8 (DP <= 32) or 4 (DP = 64) slots, shared by the other modes
    static constexpr int TMAX = 128;                   // observations per item on the DMA path
    static constexpr int STAGE_B = 2 * TMAX * 4 + TMAX * 8;   // ids of up to 2 other modes + values
    // per-wave LDS (doubles): the finishing area [img | fb | piv] aliases the gather area [ring | stage]
    static constexpr int FIN_D = DP * LD + 64 + 64;
    static constexpr int GAT_D = (8192 + STAGE_B) / 8;
    static constexpr int WAVE_LDS = FIN_D > GAT_D ? FIN_D : GAT_D;
};


// ---- accumulate one item, register path (any D, per-observation baselines): acc (MFMA C layout, lower block-triangle)
// and bred (the item's part of b) ------------------------------------------------------------------------------------
// Software pipeline over "trips" of 4*KS observations: the other-mode ids and values of trip t+2 and the gathered
// factor rows of trip t+1 are in flight while the MFMAs of trip t issue.  Lane (j = l & 15, h = l >> 4) handles
// observations h, h+4, h+8, ... of the item and elements 16 I + j of their factor rows (index-reversed).
template <int DP, int NO>
__device__ inline void accumulate_reg(const SampleArgs &a, const Item &it, int lane, d4 (&acc)[Geo<DP>::NB],
                                  double (&bred)[Geo<DP>::DB])
{
    constexpr int DB = Geo<DP>::DB, NB = Geo<DP>::NB;
    constexpr int KS = (NO == 1) ? 4 : 2;                 // k-steps (of 4 observations) per trip
    const TermDev &T = a.t[it.term];
    const int D = a.D;
    const int j = lane & 15, h = lane >> 4;
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    double bpart[DB];
    int ec[DB];                               // natural element index of reversed element 16*I + j (negative: padding)
#pragma unroll
    for (int I = 0; I < DB; I++) { bpart[I] = 0.0; ec[I] = D - 1 - (16 * I + j); }
    const int n = it.count;
    const int ntrips = (n + 4 * KS - 1) / (4 * KS);
    const int64_t qb = it.q_begin;

    int32_t ix_n[KS][NO], ix_nn[KS][NO];
    double rr_n[KS], rr_nn[KS];
    double w_n[KS][NO][DB];

#define LOAD_IDX(t, IX, RR)                                                                     \
    _Pragma("unroll") for (int k = 0; k < KS; k++) {                                            \
        const int o = (t) * 4 * KS + 4 * k + h;                                                 \
        const bool valid = o < n;                                                               \
        const int64_t q = qb + (valid ? o : 0);                                                 \
        _Pragma("unroll") for (int m = 0; m < NO; m++) IX[k][m] = T.colidx[(int64_t)m * T.nnz + q]; \
        const double base = T.linear ? T.linear[T.perm[q]] : T.mean;                            \
        RR[k] = valid ? T.vals[q] - base : 0.0;                                                 \
    }
#define LOAD_DATA(t, IX)                                                                        \
    _Pragma("unroll") for (int k = 0; k < KS; k++) {                                            \
        const bool valid = (t) * 4 * KS + 4 * k + h < n;                                        \
        _Pragma("unroll") for (int m = 0; m < NO; m++) {                                        \
            const double *f = T.fac[m] + (int64_t)IX[k][m] * D;                                 \
            _Pragma("unroll") for (int I = 0; I < DB; I++)                                      \
                w_n[k][m][I] = (valid && ec[I] >= 0) ? f[ec[I]] : 0.0;                          \
        }                                                                                       \
    }

    if (ntrips > 0) {
        LOAD_IDX(0, ix_n, rr_n)
        if (ntrips > 1) { LOAD_IDX(1, ix_nn, rr_nn) }
        LOAD_DATA(0, ix_n)
    }
    for (int t = 0; t < ntrips; t++) {
        double w_c[KS][DB], rr_c[KS];
#pragma unroll
        for (int k = 0; k < KS; k++) {
            rr_c[k] = rr_n[k];
#pragma unroll
            for (int I = 0; I < DB; I++) {
                double v = w_n[k][0][I];
#pragma unroll
                for (int m = 1; m < NO; m++) v *= w_n[k][m][I];      // Hadamard product (sampling.jl:225-227, 277-280)
                w_c[k][I] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < KS; k++) {
            rr_n[k] = rr_nn[k];
#pragma unroll
            for (int m = 0; m < NO; m++) ix_n[k][m] = ix_nn[k][m];
        }
        if (t + 1 < ntrips) { LOAD_DATA(t + 1, ix_n) }
        if (t + 2 < ntrips) { LOAD_IDX(t + 2, ix_nn, rr_nn) }
#pragma unroll
        for (int k = 0; k < KS; k++) {
            int b = 0;
#pragma unroll
            for (int I = 0; I < DB; I++) {
#pragma unroll
                for (int J = 0; J <= I; J++) {
                    acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(w_c[k][I], w_c[k][J], acc[b], 0, 0, 0);
                    b++;
                }
                bpart[I] = fma(w_c[k][I], rr_c[k], bpart[I]);
            }
        }
    }
#undef LOAD_IDX
#undef LOAD_DATA
    // scale by alpha; reduce b over the four observation groups (lanes j, j+16, j+32, j+48)
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] *= T.alpha;
#pragma unroll
    for (int I = 0; I < DB; I++) {
        double v = bpart[I] * T.alpha;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        bred[I] = v;
    }
}

// ---- accumulate one item, LDS-DMA path (even D, shared baseline), in passes of at most TMAX observations -----------
// The item's other-mode ids and values are staged in LDS by DMA once; then a ring of RING_SLOTS k-step slots is kept
// full by global_load_lds gathers (per-lane source address = chunk (l % LPR) of factor row ids[l / LPR], lane-linear LDS
// destination), P = slots/NO - 1 k-steps ahead of the MFMAs.  Nothing but DMAs uses the vector-memory counter inside
// the loop, so the waits are exact: `s_waitcnt vmcnt((P-1) * NO * IPK)` retires precisely the oldest k-step.
template <int DP, int NO>
__device__ inline void accumulate_dma(const SampleArgs &a, const Item &it, int lane, double *wl, d4 (&acc)[Geo<DP>::NB],
                                      double (&bred)[Geo<DP>::DB])
{
    using GG = Geo<DP>;
    constexpr int DB = GG::DB, NB = GG::NB, LPR = GG::LPR, ROWB = GG::ROWB, IPK = GG::IPK, SLOTB = GG::SLOTB;
    constexpr int R = GG::RING_SLOTS / NO;                // ring depth in k-steps
    constexpr int P = R - 1;                              // k-steps in flight
    static_assert(R >= 2, "ring too small");
    constexpr int TMAX = GG::TMAX;
    const TermDev &T = a.t[it.term];
    const int D = a.D;
    const int j = lane & 15, h = lane >> 4;
    char *ring = (char *)wl;
    int32_t *sidx = (int32_t *)(ring + 8192);             // [NO][TMAX]
    double *svals = (double *)(ring + 8192 + 2 * TMAX * 4);
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    double bpart[DB];
    int eoff[DB];                                         // byte offset in a gathered row of reversed element 16 I + j
    bool eok[DB];
#pragma unroll
    for (int I = 0; I < DB; I++) {
        bpart[I] = 0.0;
        const int ec = D - 1 - (16 * I + j);
        eok[I] = ec >= 0;
        eoff[I] = (ec >= 0 ? ec : 0) * 8;
    }
    const int rowbytes = D * 8;
    const int chunk = lane % LPR;
    const int coff = (chunk * 16 < rowbytes) ? chunk * 16 : rowbytes - 16;   // lanes past the row re-read its last chunk
    for (int pass0 = 0; pass0 < it.count; pass0 += TMAX) {
    const int n = (it.count - pass0 < TMAX) ? it.count - pass0 : TMAX;       // observations of this pass
    const int64_t qb = it.q_begin + pass0;
    // ---- stage ids (4 bytes per lane) and values (as dwords) of the pass
    {
        const int last = n - 1;
#pragma unroll
        for (int m = 0; m < NO; m++)
#pragma unroll
            for (int part = 0; part < TMAX / 64; part++) {
                const int o = part * 64 + lane;
                const int32_t *src = T.colidx + (int64_t)m * T.nnz + qb + (o < n ? o : last);
                __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(sidx + m * TMAX + part * 64), 4, 0, 0);
            }
#pragma unroll
        for (int part = 0; part < 2 * TMAX / 64; part++) {
            const int w = part * 64 + lane;               // dword index into the values
            const int32_t *src = (const int32_t *)(T.vals + qb) + (w < 2 * n ? w : 2 * last + (w & 1));
            __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)((int32_t *)svals + part * 64), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int nks = (n + 3) >> 2;

    auto issue = [&](int ks) {
        // gather the 4 factor rows of k-step ks (clamped into the pass) of every other mode into slot ks % R
        char *slot = ring + (ks % R) * (NO * SLOTB);
#pragma unroll
        for (int m = 0; m < NO; m++)
#pragma unroll
            for (int q = 0; q < IPK; q++) {
                int o = 4 * ks + q * (64 / LPR) + lane / LPR;
                o = o < n ? o : n - 1;
                const char *src = (const char *)(T.fac[m] + (int64_t)sidx[m * TMAX + o] * D) + coff;
                __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(slot + m * SLOTB + q * 1024), 16, 0, 0);
            }
    };
#pragma unroll
    for (int s = 0; s < P; s++) issue(s);
    for (int ks = 0; ks < nks; ks++) {
        // exactly P k-steps are outstanding here: ks .. ks+P-1; retire the oldest
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * NO * IPK) : "memory");
        const char *slot = ring + (ks % R) * (NO * SLOTB) + h * ROWB;
        double w[DB];
#pragma unroll
        for (int I = 0; I < DB; I++) {
            double v = *(const double *)(slot + eoff[I]);
#pragma unroll
            for (int m = 1; m < NO; m++) v *= *(const double *)(slot + m * SLOTB + eoff[I]);    // Hadamard product
            w[I] = v;
        }
        const int o = 4 * ks + h;
        const bool valid = o < n;
        const double rr = valid ? svals[valid ? o : 0] - T.mean : 0.0;
#pragma unroll
        for (int I = 0; I < DB; I++) w[I] = (valid && eok[I]) ? w[I] : 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // operands are in registers before their slot is reused
        issue(ks + P);
        int b = 0;
#pragma unroll
        for (int I = 0; I < DB; I++) {
#pragma unroll
            for (int J = 0; J <= I; J++) {
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[I], w[J], acc[b], 0, 0, 0);
                b++;
            }
            bpart[I] = fma(w[I], rr, bpart[I]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the run-ahead gathers before the stage is rewritten
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the run-ahead gathers before the LDS is reused
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] *= T.alpha;
#pragma unroll
    for (int I = 0; I < DB; I++) {
        double v = bpart[I] * T.alpha;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        bred[I] = v;
    }
}

// path and other-mode count are wave-uniform
template <int DP>
__device__ inline void accumulate_any(const SampleArgs &a, const Item &it, int lane, double *wl, d4 (&acc)[Geo<DP>::NB],
                                      double (&bred)[Geo<DP>::DB])
{
    const TermDev &T = a.t[it.term];
    const int no = T.n_other;
    const bool dma = (a.D % 2 == 0) && T.linear == nullptr;
    if (dma && no == 1) {
        accumulate_dma<DP, 1>(a, it, lane, wl, acc, bred);
    } else if (dma && no == 2) {
        accumulate_dma<DP, 2>(a, it, lane, wl, acc, bred);
    } else {
        if (no == 1) accumulate_reg<DP, 1>(a, it, lane, acc, bred);
        else if (no == 2) accumulate_reg<DP, 2>(a, it, lane, acc, bred);
        else accumulate_reg<DP, 3>(a, it, lane, acc, bred);
    }
}

// ---- spread an accumulator (C layout) as a full symmetric image P~[i][c] at img[i*LD + c] ----------------------------
// C layout of block (I,J): lane l, register r holds element (row 16I + (l>>4) + 4r, column 16J + (l&15)).
template <int DP>
__device__ inline void acc_to_image(const d4 (&acc)[Geo<DP>::NB], double *img, int lane)
{
    constexpr int DB = Geo<DP>::DB, LD = Geo<DP>::LD;
    const int j = lane & 15, h = lane >> 4;
    int b = 0;
#pragma unroll
    for (int I = 0; I < DB; I++)
#pragma unroll
        for (int J = 0; J <= I; J++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * I + h + 4 * r, colm = 16 * J + j;
                img[row * LD + colm] = acc[b][r];
                if (I != J) img[colm * LD + row] = acc[b][r];
            }
            b++;
        }
}

// ---- finish G rows at once: lane group grp = lane / DP owns row myrow; col[] = column c of P~ (prior included) --------
template <int DP, bool DUMP>
__device__ inline void finish_rows(const SampleArgs &a, int64_t myrow, double (&col)[DP], double bj, double *wl, int lane)
{
    constexpr int LD = Geo<DP>::LD;
    const int D = a.D;
    const int c = lane % DP;
    const int ec = D - 1 - c;
    double *img = wl;                       // DP x LD image (conversion / transposition buffer)
    double *fb = wl + DP * LD;              // [64] broadcast row
    double *piv = fb + 64;                  // [64] 1 / pivot

    if (DUMP) {
        if (myrow >= 0 && ec >= 0) {
#pragma unroll
            for (int i = 0; i < DP; i++) {
                const int ei = D - 1 - i;
                if (ei >= 0) a.P_dump[(myrow * D + ec) * D + ei] = col[i];
            }
            a.b_dump[myrow * D + ec] = bj;
        }
        return;
    }
    double p_own;
    if (wl_factor<DP>(col, p_own, fb, piv, lane) && myrow >= 0) atomicOr(a.flag, 1);
    const double sq_own = p_own * fast_rsqrt(p_own);                 // L[c][c] = sqrt(p_c)
    // L w = b, y = w + z, carried as yh = y sqrt(p) = b' + z sqrt(p)   (z reversed: column c takes normal number D-1-c)
    const double bp = wl_forward<DP>(col, bj, piv, lane);
    double yh = 0.0;
    if (myrow >= 0 && ec >= 0)
        yh = fma(bdf_normal(a.seed, *a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)myrow, ec), sq_own, bp);
    wl_transpose<DP>(col, img, lane);
    const double x = wl_backward<DP>(col, yh, piv, lane);            // L' x = y
    if (myrow >= 0 && ec >= 0) a.out[myrow * D + ec] = x;
}

// column c of the prior in reversed coordinates: Lambda~[i][c] (identity padding) and (Lambda mu_i)~[c]
template <int DP>
__device__ inline void add_prior(const SampleArgs &a, int64_t myrow, int c, double (&col)[DP], double &bj)
{
    const int D = a.D;
    const int ec = D - 1 - c;
    if (ec >= 0 && myrow >= 0) {
        const double *mu_i = a.mu_is_matrix ? a.mu + myrow * D : a.mu;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < DP; i++) {
            const int ei = D - 1 - i;
            if (ei >= 0) {
                const double lam = a.Lambda[ei + (int64_t)ec * D];
                col[i] += lam;
                s = fma(lam, mu_i[ei], s);        // (Lambda mu)[ec] = sum_ei Lambda[ec][ei] mu[ei] (Lambda symmetric)
            }
        }
        bj += s;
    } else {
#pragma unroll
        for (int i = 0; i < DP; i++) col[i] = (i == c) ? 1.0 : 0.0;
        bj = 0.0;
    }
}

// ---- launch 1: accumulate every item; finish the direct rows -------------------------------------------------------
template <int DP, bool DUMP>
__global__ __launch_bounds__(64 * Geo<DP>::WPB, (DP == 64) ? 1 : BDF_K1_WAVES) void k_rows_accum(SampleArgs a, PlanDev p)
{
    constexpr int G = Geo<DP>::G, DB = Geo<DP>::DB, NB = Geo<DP>::NB, LD = Geo<DP>::LD, PSZ = Geo<DP>::PSZ;
    constexpr int WPB = Geo<DP>::WPB;
    __shared__ __attribute__((aligned(16))) double lds[WPB * Geo<DP>::WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *wl = lds + wave * Geo<DP>::WAVE_LDS;
    const int64_t wid = (int64_t)blockIdx.x * WPB + wave;

    const int64_t n_dwaves = p.n_direct / G;
    if (wid >= n_dwaves) {
        // a split item (launched after the direct rows, so that the short items fill the tail of the launch):
        // partial to the slab, slot layout [block*4 + r][lane] then b[I][j]
        if (wid - n_dwaves >= p.n_split) return;
        d4 acc[NB];
        double bred[DB];
        const Item it = p.split[wid - n_dwaves];
        accumulate_any<DP>(a, it, lane, wl, acc, bred);
        double *dst = p.partials + (int64_t)it.slot * PSZ;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(b * 4 + r) * 64 + lane] = acc[b][r];
        if (lane < 16) {
#pragma unroll
            for (int I = 0; I < DB; I++) dst[NB * 4 * 64 + I * 16 + lane] = bred[I];
        }
        return;
    }
    const int64_t first = wid * G;
    const int grp = lane / DP, c = lane % DP;
    // accumulate the G rows of this wave one after the other; their accumulators stay in registers (NB*4 doubles
    // each) until all are done, so that the column array of the finishing phase is not live during the gathers
    d4 accg[G][NB];
    double bredg[G][DB];
    int64_t rows[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const Item it = p.direct[first + g];
        rows[g] = it.row;
#pragma unroll
        for (int b = 0; b < NB; b++) accg[g][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int I = 0; I < DB; I++) bredg[g][I] = 0.0;
        if (it.row >= 0 && it.count > 0) accumulate_any<DP>(a, it, lane, wl, accg[g], bredg[g]);    // wave-uniform branch
    }
    double col[DP];
    double bj = 0.0;
    int64_t myrow = -1;
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (rows[g] >= 0) {
            wave_sync();
            acc_to_image<DP>(accg[g], wl, lane);
            if (lane < 16) {
#pragma unroll
                for (int I = 0; I < DB; I++) wl[DP * LD + I * 16 + lane] = bredg[g][I];
            }
            wave_sync();
            if (grp == g) {
#pragma unroll
                for (int i = 0; i < DP; i++) col[i] = wl[i * LD + c];
                bj = wl[DP * LD + c];
                myrow = rows[g];
            }
        }
    }
    wave_sync();
    add_prior<DP>(a, myrow, c, col, bj);
    finish_rows<DP, DUMP>(a, myrow, col, bj, wl, lane);
}

// ---- items-only launch (every row split): accumulate items to the slab, nothing else -------------------------------------
template <int DP>
__global__ __launch_bounds__(64 * Geo<DP>::WPB) void k_rows_items(SampleArgs a, PlanDev p)
{
    constexpr int DB = Geo<DP>::DB, NB = Geo<DP>::NB, PSZ = Geo<DP>::PSZ, WPB = Geo<DP>::WPB;
    __shared__ __attribute__((aligned(16))) double lds[WPB * Geo<DP>::GAT_D];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *wl = lds + wave * Geo<DP>::GAT_D;
    const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
    if (wid >= p.n_split) return;
    d4 acc[NB];
    double bred[DB];
    const Item it = p.split[wid];
    accumulate_any<DP>(a, it, lane, wl, acc, bred);
    double *dst = p.partials + (int64_t)it.slot * PSZ;
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) dst[(b * 4 + r) * 64 + lane] = acc[b][r];
    if (lane < 16) {
#pragma unroll
        for (int I = 0; I < DB; I++) dst[NB * 4 * 64 + I * 16 + lane] = bred[I];
    }
}

// ---- launch 2: add the partials of the split rows in slot order and finish them --------------------------------------
template <int DP, bool DUMP>
__global__ __launch_bounds__(64 * Geo<DP>::WPB, (DP == 64) ? 1 : BDF_K1_WAVES) void k_rows_finish(SampleArgs a, PlanDev p)
{
    constexpr int G = Geo<DP>::G, DB = Geo<DP>::DB, NB = Geo<DP>::NB, LD = Geo<DP>::LD, PSZ = Geo<DP>::PSZ;
    constexpr int WPB = Geo<DP>::WPB;
    __shared__ __attribute__((aligned(16))) double lds[WPB * Geo<DP>::WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *wl = lds + wave * Geo<DP>::WAVE_LDS;
    const int64_t first = ((int64_t)blockIdx.x * WPB + wave) * G;
    if (first >= p.n_split_rows) return;
    const int grp = lane / DP, c = lane % DP;
    double col[DP];
    double bj = 0.0;
    int64_t myrow = -1;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const SplitRow sr = p.rows[first + g];
        if (sr.row >= 0) {
            d4 acc[NB];
            double bred[DB];
#pragma unroll
            for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < DB; I++) bred[I] = 0.0;
            // slot order is fixed; four slots are loaded per trip so that their latencies overlap
            for (int s0 = 0; s0 < sr.n_slots; s0 += 4) {
                double v[4][NB * 4 + DB];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int s = (s0 + u < sr.n_slots) ? s0 + u : sr.n_slots - 1;
                    const double *src = p.partials + (int64_t)(sr.slot_begin + s) * PSZ;
#pragma unroll
                    for (int e = 0; e < NB * 4; e++) v[u][e] = src[e * 64 + lane];
#pragma unroll
                    for (int I = 0; I < DB; I++) v[u][NB * 4 + I] = src[NB * 4 * 64 + I * 16 + (lane & 15)];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (s0 + u < sr.n_slots) {
#pragma unroll
                        for (int b = 0; b < NB; b++)
#pragma unroll
                            for (int r = 0; r < 4; r++) acc[b][r] += v[u][b * 4 + r];
#pragma unroll
                        for (int I = 0; I < DB; I++) bred[I] += v[u][NB * 4 + I];
                    }
                }
            }
            wave_sync();
            acc_to_image<DP>(acc, wl, lane);
            if (lane < 16) {
#pragma unroll
                for (int I = 0; I < DB; I++) wl[DP * LD + I * 16 + lane] = bred[I];
            }
            wave_sync();
            if (grp == g) {
#pragma unroll
                for (int i = 0; i < DP; i++) col[i] = wl[i * LD + c];
                bj = wl[DP * LD + c];
                myrow = sr.row;
            }
        }
    }
    wave_sync();
    add_prior<DP>(a, myrow, c, col, bj);
    finish_rows<DP, DUMP>(a, myrow, col, bj, wl, lane);
}

// ---- host: the plan (items, split rows, slab) for a (terms, row list) combination, cached per context ---------------
struct PlanKey {
    uint64_t rel[BDF_MAX_TERMS];      // relation serials
    int mode[BDF_MAX_TERMS];
    int n_terms, DP, T;
    int shard, n_shards;
    bool operator<(const PlanKey &o) const { return memcmp(this, &o, sizeof(PlanKey)) < 0; }
};

struct Plan {
    PlanDev dev;
    Item *direct_dev = nullptr, *split_dev = nullptr;
    SplitRow *rows_dev = nullptr;
    double *partials_dev = nullptr;
};

struct PlanCache {
    std::map<PlanKey, Plan> plans;
};

std::mutex g_cache_mutex;
std::map<bdf_ctx *, PlanCache> g_caches;

template <typename T>
int to_device(const std::vector<T> &v, T **out)
{
    BDF_HIP(hipMalloc((void **)out, std::max<size_t>(v.size() * sizeof(T), 8)));
    if (!v.empty()) BDF_HIP(hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return BDF_OK;
}

int build_plan(bdf_ctx *ctx, const PlanKey &key, const bdf_rel *const *rels, const std::vector<int32_t> &rows, int G,
               int psz, Plan &plan)
{
    const int T = key.T;
    std::vector<Item> direct, split;
    std::vector<SplitRow> srows;
    for (int32_t row : rows) {
        int n_items = 0;
        for (int r = 0; r < key.n_terms; r++) {
            const auto &rp = rels[r]->idx[key.mode[r]].rowptr;
            const int64_t n = rp[(size_t)row + 1] - rp[(size_t)row];
            n_items += (int)((n + T - 1) / T);
        }
        static const bool all_split = getenv("BDF_ALL_SPLIT") != nullptr;
        if (n_items <= 1 && !all_split) {
            Item it{row, 0, 0, 0, -1};
            for (int r = 0; r < key.n_terms; r++) {
                const auto &rp = rels[r]->idx[key.mode[r]].rowptr;
                const int64_t n = rp[(size_t)row + 1] - rp[(size_t)row];
                if (n > 0) { it.term = r; it.q_begin = rp[(size_t)row]; it.count = (int32_t)n; }
            }
            direct.push_back(it);
        } else {
            if (n_items == 0) {          // all-split mode, empty row: one empty item so that the row has a slot
                split.push_back(Item{row, 0, 0, 0, (int32_t)split.size()});
                srows.push_back(SplitRow{row, (int32_t)split.size() - 1, 1, 0});
                continue;
            }
            SplitRow sr{row, (int32_t)split.size(), n_items, 0};
            for (int r = 0; r < key.n_terms; r++) {
                const auto &rp = rels[r]->idx[key.mode[r]].rowptr;
                const int64_t beg = rp[(size_t)row], n = rp[(size_t)row + 1] - beg;
                const int pieces = (int)((n + T - 1) / T);
                for (int s = 0; s < pieces; s++) {
                    // equal pieces rather than T, T, ..., remainder
                    const int64_t b0 = beg + n * s / pieces, b1 = beg + n * (s + 1) / pieces;
                    split.push_back(Item{row, r, b0, (int32_t)(b1 - b0), (int32_t)split.size()});
                }
            }
            srows.push_back(sr);
        }
    }
    while (direct.size() % (size_t)G) direct.push_back(Item{-1, 0, 0, 0, -1});
    while (srows.size() % (size_t)G) srows.push_back(SplitRow{-1, 0, 0, 0});
    int rc;
    if ((rc = to_device(direct, &plan.direct_dev)) || (rc = to_device(split, &plan.split_dev)) ||
        (rc = to_device(srows, &plan.rows_dev)))
        return rc;
    BDF_HIP(hipMalloc((void **)&plan.partials_dev, std::max<size_t>(split.size() * (size_t)psz * sizeof(double), 8)));
    plan.dev.direct = plan.direct_dev; plan.dev.n_direct = (int32_t)direct.size();
    plan.dev.split = plan.split_dev;   plan.dev.n_split = (int32_t)split.size();
    plan.dev.rows = plan.rows_dev;     plan.dev.n_split_rows = (int32_t)srows.size();
    plan.dev.partials = plan.partials_dev;
    return BDF_OK;
}

template <int DP>
int launch(bdf_ctx *ctx, const SampleArgs &a, const PlanDev &p, bool dump)
{
    constexpr int G = Geo<DP>::G, WPB = Geo<DP>::WPB;
    const int64_t waves1 = (int64_t)p.n_split + p.n_direct / G;
    if (p.n_direct == 0 && p.n_split > 0) {
        dim3 grid((unsigned)((p.n_split + WPB - 1) / WPB)), block(64 * WPB);
        hipLaunchKernelGGL((k_rows_items<DP>), grid, block, 0, ctx->stream, a, p);
        BDF_HIP(hipGetLastError());
    } else if (waves1 > 0) {
        dim3 grid((unsigned)((waves1 + WPB - 1) / WPB)), block(64 * WPB);
        if (dump) hipLaunchKernelGGL((k_rows_accum<DP, true>), grid, block, 0, ctx->stream, a, p);
        else      hipLaunchKernelGGL((k_rows_accum<DP, false>), grid, block, 0, ctx->stream, a, p);
        BDF_HIP(hipGetLastError());
    }
    const int64_t waves2 = p.n_split_rows / G;
    if (waves2 > 0) {
        dim3 grid((unsigned)((waves2 + WPB - 1) / WPB)), block(64 * WPB);
        if (dump) hipLaunchKernelGGL((k_rows_finish<DP, true>), grid, block, 0, ctx->stream, a, p);
        else      hipLaunchKernelGGL((k_rows_finish<DP, false>), grid, block, 0, ctx->stream, a, p);
        BDF_HIP(hipGetLastError());
    }
    return BDF_OK;
}

}  // namespace

void bdf_plans_release(bdf_ctx *ctx, uint64_t rel_serial)
{
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto it = g_caches.find(ctx);
    if (it == g_caches.end()) return;
    auto &plans = it->second.plans;
    for (auto kv = plans.begin(); kv != plans.end();) {
        bool hit = rel_serial == 0;
        for (int r = 0; r < kv->first.n_terms; r++) hit = hit || kv->first.rel[r] == rel_serial;
        if (hit) {
            hipFree(kv->second.direct_dev); hipFree(kv->second.split_dev); hipFree(kv->second.rows_dev);
            hipFree(kv->second.partials_dev);
            kv = plans.erase(kv);
        } else {
            ++kv;
        }
    }
    if (rel_serial == 0) g_caches.erase(it);
}

int bdf_launch_sample_rows(bdf_ctx *ctx, const SampleArgs &a, const bdf_rel *const *rels, const int *modes, int shard,
                           int n_shards, bool dump)
{
    const int DP = a.D <= 16 ? 16 : (a.D <= 32 ? 32 : 64);
    const int G = 64 / DP;
    const int DB = DP / 16, NB = DB * (DB + 1) / 2;
    const int psz = NB * 4 * 64 + DB * 16;
    PlanKey key;
    memset(&key, 0, sizeof(key));
    for (int r = 0; r < a.n_terms; r++) { key.rel[r] = rels[r]->serial; key.mode[r] = modes[r]; }
    key.n_terms = a.n_terms; key.DP = DP; key.T = ctx->item_size; key.shard = shard; key.n_shards = n_shards;

    Plan *plan;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        PlanCache &cache = g_caches[ctx];
        auto it = cache.plans.find(key);
        if (it == cache.plans.end()) {
            // rows of this shard: positions shard, shard + n_shards, ... of the degree-descending order of the first
            // relation (the reference deals rows i:P:N to its P workers for the same balance, sampling.jl:154)
            const std::vector<int32_t> &order = rels[0]->idx[modes[0]].order;
            std::vector<int32_t> rows;
            for (size_t pos = (size_t)shard; pos < order.size(); pos += (size_t)n_shards) rows.push_back(order[pos]);
            Plan np;
            int rc = build_plan(ctx, key, rels, rows, G, psz, np);
            if (rc) return rc;
            it = cache.plans.emplace(key, np).first;
        }
        plan = &it->second;
    }
    if (DP == 16) return launch<16>(ctx, a, plan->dev, dump);
    if (DP == 32) return launch<32>(ctx, a, plan->dev, dump);
    return launch<64>(ctx, a, plan->dev, dump);
}
