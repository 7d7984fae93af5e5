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
//     scratch slab and count themselves in; the wave whose arrival completes the row adds the row's partials in slot
//     order and finishes it, inside the same launch.
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

#ifndef BDF_K1_WPB
#define BDF_K1_WPB 4
#endif
#ifndef BDF_K1_WAVES
#define BDF_K1_WAVES 3
#endif
#ifndef BDF_K1_RING_B
#define BDF_K1_RING_B 8192        // bytes of the gather ring per wave (D <= 32)
#endif
#ifndef BDF_K1_TMAX
#define BDF_K1_TMAX 192           // observations staged per pass
#endif

#ifdef BDF_K1_STAMPS
#define STAMP(slot) do { if (lane == 0 && a.b_dump) ((unsigned long long *)a.b_dump)[wid * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(slot) do { } while (0)
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
    int32_t srow;         // index of the row in the split-row table (split items)
    int32_t _pad;
};

struct SplitRow {
    int32_t row;
    int32_t slot_begin, n_slots;
    int32_t _pad;
};

struct PlanDev {
    const Item *direct;   int32_t n_direct;      // padded to a multiple of G
    const Item *split;    int32_t n_split;
    const SplitRow *rows; int32_t n_split_rows;
    double *partials;                            // n_split * PSZ doubles
    int32_t *arrived;                            // per split row: items that have published their partial (self-resetting)
};

template <int DP>
struct Geo {
    static constexpr int G = 64 / DP;                  // rows finished together by one wave
    static constexpr int DB = DP / 16;                 // 16-wide blocks per dimension
    static constexpr int NB = DB * (DB + 1) / 2;       // lower block-triangle
    static constexpr int LD = DP + 1;                  // padded leading dimension of the LDS images
    static constexpr int PSZ = NB * 4 * 64 + DB * 16;  // doubles per partial slot
    static constexpr int WPB = (DP == 64) ? 1 : BDF_K1_WPB;   // waves per workgroup (static LDS must stay under 64 KB)
    // LDS-DMA gather ring: a slot holds the 4 gathered factor rows of one MFMA k-step
    static constexpr int LPR = (DP == 64) ? 32 : 16;   // lanes (16 bytes each) per gathered row
    static constexpr int ROWB = LPR * 16;              // bytes between rows in a slot
    static constexpr int IPK = 4 * LPR / 64;           // DMA instructions per k-step and other mode
    static constexpr int SLOTB = 4 * ROWB;             // bytes per slot
    static constexpr int RING_B = (DP == 64) ? 8192 : BDF_K1_RING_B;
    static constexpr int RING_SLOTS = RING_B / SLOTB;  // 8 (DP <= 32) or 4 (DP = 64) slots, shared by the other modes
    static constexpr int TMAX = BDF_K1_TMAX;                   // observations staged per pass on the DMA path
    // stage of one item: [values TMAX x 8][ids of other mode 0, TMAX x 4][ids of other mode 1, TMAX x 4]; consecutive
    // stages are STAGE_B apart, so an item with two other modes needs the room of the following stage as well
    static constexpr int STAGE_B = TMAX * 12;
    static constexpr int NSTAGE = (G >= 2) ? 2 : 1;    // items of a wave staged ahead (matrix relations only)
    // per-wave LDS (doubles): the finishing area [img | fb | piv] aliases the gather area [ring | stage]
    static constexpr int FIN_D = (DP * LD + 64 > G * (DP * (DP + 1) / 2) + 64) ? DP * LD + 64 : G * (DP * (DP + 1) / 2) + 64;
    static constexpr int GAT_D = (RING_B + (NSTAGE >= 2 ? NSTAGE * STAGE_B : STAGE_B + TMAX * 4)) / 8;
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

// ---- accumulate one item, LDS-DMA path (even D, shared baseline) --------------------------------------------------------
// The item's other-mode ids and values are staged in LDS by DMA (stage_item, up to TMAX observations per pass); then a
// ring of k-step slots is kept full by global_load_lds gathers (per-lane source address = chunk (l % LPR) of factor row
// ids[l / LPR], lane-linear LDS destination), P = R - 1 k-steps ahead of the MFMAs.  Nothing but DMAs uses the
// vector-memory counter inside the loop, so the waits are exact: `s_waitcnt vmcnt((P-1) * NO * IPK)` retires precisely
// the oldest k-step.
template <int DP, int NO>
__device__ inline void stage_item(const TermDev &T, int64_t qb, int n, int lane, char *stage)
{
    constexpr int TMAX = Geo<DP>::TMAX;
    int32_t *svals = (int32_t *)stage;                    // TMAX doubles as dwords
    int32_t *sidx = (int32_t *)(stage + TMAX * 8);        // [NO][TMAX]
    const int last = n - 1;
#pragma unroll
    for (int m = 0; m < NO; m++)
#pragma unroll
        for (int part = 0; part < (TMAX + 63) / 64; part++) {
            const int o = part * 64 + lane;
            const int32_t *src = T.colidx + (int64_t)m * T.nnz + qb + (o < n ? o : last);
            if (part * 64 < TMAX && (part + 1) * 64 <= TMAX + 63)
                __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(sidx + m * TMAX + part * 64), 4, 0, 0);
        }
#pragma unroll
    for (int part = 0; part < (2 * TMAX + 63) / 64; part++) {
        const int w = part * 64 + lane;                   // dword index into the values
        const int32_t *src = (const int32_t *)(T.vals + qb) + (w < 2 * n ? w : 2 * last + (w & 1));
        __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(svals + part * 64), 4, 0, 0);
    }
}

template <int DP, int NO>
__device__ inline void accumulate_dma(const SampleArgs &a, const Item &it, int lane, double *wl, int stage_sel,
                                      bool prestaged, d4 (&acc)[Geo<DP>::NB], double (&bred)[Geo<DP>::DB])
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
    char *stage = ring + GG::RING_B + stage_sel * GG::STAGE_B;
    const double *svals = (const double *)stage;
    const int32_t *sidx = (const int32_t *)(stage + TMAX * 8);
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
    const int lrow = lane / LPR;
    for (int pass0 = 0; pass0 < it.count; pass0 += TMAX) {
        const int n = (it.count - pass0 < TMAX) ? it.count - pass0 : TMAX;   // observations of this pass
        if (!(prestaged && pass0 == 0)) stage_item<DP, NO>(T, it.q_begin + pass0, n, lane, stage);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int nks = (n + 3) >> 2;

        // gather the 4 factor rows of a k-step into slot ks % R; rix[m][q] = row index read from the stage
        auto gather = [&](int ks, const int32_t (&rix)[NO][IPK]) {
            char *slot = ring + (ks % R) * (NO * SLOTB);
#pragma unroll
            for (int m = 0; m < NO; m++)
#pragma unroll
                for (int q = 0; q < IPK; q++) {
                    const char *src = (const char *)(T.fac[m] + (int64_t)rix[m][q] * D) + coff;
                    __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(slot + m * SLOTB + q * 1024), 16, 0, 0);
                }
        };
        auto read_idx = [&](int ks, int32_t (&rix)[NO][IPK]) {
#pragma unroll
            for (int m = 0; m < NO; m++)
#pragma unroll
                for (int q = 0; q < IPK; q++) {
                    int o = 4 * ks + q * (64 / LPR) + lrow;
                    o = o < n ? o : n - 1;                   // past the end: any row of the pass (operands are zeroed)
                    rix[m][q] = sidx[m * TMAX + o];
                }
        };
#pragma unroll
        for (int s = 0; s < P; s++) {
            int32_t rix[NO][IPK];
            read_idx(s, rix);
            gather(s, rix);
        }
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
            int32_t rix[NO][IPK];
            read_idx(ks + P, rix);                           // same LDS round trip as the operands
#pragma unroll
            for (int I = 0; I < DB; I++) w[I] = (valid && eok[I]) ? w[I] : 0.0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // operands are in registers before their slot is reused
            gather(ks + P, rix);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the run-ahead gathers before the stage is rewritten
    }
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

template <int DP>
__device__ inline bool item_uses_dma(const SampleArgs &a, const Item &it)
{
    const TermDev &T = a.t[it.term];
    return (a.D % 2 == 0) && T.linear == nullptr && T.n_other <= 2;
}

// issue the staging DMAs of an item ahead of its accumulation (returns false if the item takes the register path)
template <int DP>
__device__ inline bool prestage(const SampleArgs &a, const Item &it, int lane, double *wl, int stage_sel)
{
    if (it.row < 0 || it.count <= 0 || !item_uses_dma<DP>(a, it)) return false;
    const TermDev &T = a.t[it.term];
    if (T.n_other != 1 && stage_sel != 0) return false;       // a two-mode stage spills into the next one
    char *stage = (char *)wl + Geo<DP>::RING_B + stage_sel * Geo<DP>::STAGE_B;
    const int n = it.count < Geo<DP>::TMAX ? it.count : Geo<DP>::TMAX;
    if (T.n_other == 1) stage_item<DP, 1>(T, it.q_begin, n, lane, stage);
    else stage_item<DP, 2>(T, it.q_begin, n, lane, stage);
    return true;
}

// path and other-mode count are wave-uniform
template <int DP>
__device__ inline void accumulate_any(const SampleArgs &a, const Item &it, int lane, double *wl, int stage_sel,
                                      bool prestaged, d4 (&acc)[Geo<DP>::NB], double (&bred)[Geo<DP>::DB])
{
    const TermDev &T = a.t[it.term];
    const int no = T.n_other;
    if (item_uses_dma<DP>(a, it)) {
        if (no == 1) accumulate_dma<DP, 1>(a, it, lane, wl, stage_sel, prestaged, acc, bred);
        else accumulate_dma<DP, 2>(a, it, lane, wl, stage_sel, prestaged, acc, bred);   // 3 other modes: register path
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
__device__ inline void finish_rows(const SampleArgs &a, int64_t myrow, double (&col)[DP], double bj, double znorm,
                                   double *wl, int lane, int64_t wid)
{
    constexpr int LD = Geo<DP>::LD;
    const int D = a.D;
    const int c = lane % DP;
    const int ec = D - 1 - c;
    double *tri = wl;                       // packed factor (aliases the conversion image, which has been consumed)

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
    double p_own, rp_own;
    STAMP(4);
    if (wl_factor<DP>(col, p_own, rp_own, tri, lane) && myrow >= 0) atomicOr(a.flag, 1);
    STAMP(5);
    const double sq_own = p_own * fast_rsqrt(p_own);                 // L[c][c] = sqrt(p_c)
    // L w = b, y = w + z, carried as yh = y sqrt(p) = b' + z sqrt(p)   (z reversed: column c takes normal number D-1-c)
    const double bp = wl_forward<DP>(col, bj, rp_own, lane);
    const double yh = (myrow >= 0 && ec >= 0) ? fma(znorm, sq_own, bp) : 0.0;
    STAMP(6);
    STAMP(7);
    const double x = wl_backward<DP>(tri, yh, rp_own, lane);         // L' x = y
    if (myrow >= 0 && ec >= 0) a.out[myrow * D + ec] = x;
    STAMP(8);
}

// ---- prior: Lambda~ in the accumulator (C) layout, identity on the padding; loaded once per wave, added to every row's
// accumulator before the layout change.  The prior part of b, Lambda mu_i, comes from a small pre-launch (k_prior_b).
template <int DP>
__device__ inline void load_prior_c(const SampleArgs &a, int lane, d4 (&lamc)[Geo<DP>::NB])
{
    constexpr int DB = Geo<DP>::DB;
    const int D = a.D;
    const int j = lane & 15, h = lane >> 4;
    int b = 0;
#pragma unroll
    for (int I = 0; I < DB; I++)
#pragma unroll
        for (int J = 0; J <= I; J++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * I + h + 4 * r, colm = 16 * J + j;
                const int er = D - 1 - row, ecm = D - 1 - colm;
                double v = (row == colm) ? 1.0 : 0.0;
                if (er >= 0 && ecm >= 0) v = a.Lambda[er + (int64_t)ecm * D];
                else if (er >= 0 || ecm >= 0) v = 0.0;
                lamc[b][r] = v;
            }
            b++;
        }
}

__global__ __launch_bounds__(256) void k_prior_b(int D, int64_t nrows, const double *Lambda, const double *mu,
                                                 int mu_is_matrix, double *out)
{
    // out[row*D + e] = sum_i Lambda[e][i] mu_row[i]   (nrows = 1 for a shared prior mean); one wave per output
    const int lane = threadIdx.x & 63;
    const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (idx >= nrows * D) return;
    const int64_t row = idx / D;
    const int e = (int)(idx % D);
    const double *m = mu_is_matrix ? mu + row * D : mu;
    double s = (lane < D) ? Lambda[e + (int64_t)lane * D] * m[lane] : 0.0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[idx] = s;
}

// ---- sum the partials of a split row in slot order (fixed order: the result does not depend on which wave does it) ---
template <int DP>
__device__ inline void sum_partials(const PlanDev &p, const SplitRow &sr, int lane, d4 (&acc)[Geo<DP>::NB],
                                    double (&bred)[Geo<DP>::DB])
{
    constexpr int DB = Geo<DP>::DB, NB = Geo<DP>::NB, PSZ = Geo<DP>::PSZ;
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < DB; I++) bred[I] = 0.0;
    // four slots are loaded per trip so that their latencies overlap
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
}

// ---- the launch: wave w < n_split accumulates split item w and publishes its partial; the wave whose publication
// completes a row finishes that row (agent-scope release / acquire around a per-row arrival counter, placement
// independent: cdna_hip_programming.md Guideline 16).  The remaining waves take G direct rows each. -----------------------
template <int DP, bool DUMP>
__global__ __launch_bounds__(64 * Geo<DP>::WPB, (DP == 64) ? 1 : BDF_K1_WAVES) void k_rows(SampleArgs a, PlanDev p)
{
    constexpr int G = Geo<DP>::G, DB = Geo<DP>::DB, NB = Geo<DP>::NB, LD = Geo<DP>::LD, PSZ = Geo<DP>::PSZ;
    constexpr int WPB = Geo<DP>::WPB;
    __shared__ __attribute__((aligned(16))) double lds[WPB * Geo<DP>::WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *wl = lds + wave * Geo<DP>::WAVE_LDS;
    const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
    const int grp = lane / DP, c = lane % DP;
    double col[DP];
    double bj = 0.0, znorm = 0.0;
    int64_t myrow = -1;
    STAMP(0);

    if (wid < p.n_split) {
        d4 acc[NB];
        double bred[DB];
        const Item it = p.split[wid];
        if (it.count > 0) accumulate_any<DP>(a, it, lane, wl, 0, false, acc, bred);
        else {
#pragma unroll
            for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < DB; I++) bred[I] = 0.0;
        }
        // slot layout [block*4 + r][lane] then b[I][j]; write-through (sc1) stores: the slab needs no L2 write-back
        double *dst = p.partials + (int64_t)it.slot * PSZ;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                __hip_atomic_store(dst + (b * 4 + r) * 64 + lane, acc[b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < 16) {
#pragma unroll
            for (int I = 0; I < DB; I++)
                __hip_atomic_store(dst + NB * 4 * 64 + I * 16 + lane, bred[I], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // publish: every lane's write-through stores have completed, then one arrival
        const SplitRow sr = p.rows[it.srow];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(p.arrived + it.srow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        STAMP(1);
        if (old != sr.n_slots - 1) return;                      // not the last item of the row
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) p.arrived[it.srow] = 0;                  // ready for the next launch
        // the row's normal (column c takes number D-1-c of the row's stream), drawn while few registers are live
        if (grp == 0 && a.D - 1 - c >= 0 && !DUMP)
            znorm = bdf_normal(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)sr.row, a.D - 1 - c);
        sum_partials<DP>(p, sr, lane, acc, bred);
        STAMP(2);
        {
            d4 lamc[NB];
            load_prior_c<DP>(a, lane, lamc);
#pragma unroll
            for (int b = 0; b < NB; b++) acc[b] += lamc[b];
        }
        wave_sync();
        acc_to_image<DP>(acc, wl, lane);
        if (lane < 16) {
#pragma unroll
            for (int I = 0; I < DB; I++) wl[DP * LD + I * 16 + lane] = bred[I];
        }
        wave_sync();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < DP; i++) col[i] = wl[i * LD + c];
            bj = wl[DP * LD + c];
            myrow = sr.row;
        }
    } else {
        const int64_t first = (wid - p.n_split) * G;
        if (first >= p.n_direct) return;
        // accumulate the G rows of this wave one after the other; their accumulators stay in registers (NB*4 doubles
        // each) until all are done, so that the column array of the finishing phase is not live during the gathers
        d4 accg[G][NB];
        double bredg[G][DB];
        int64_t rows[G];
        bool staged[G];
        // the first NSTAGE items are staged together (one exposed latency) when they are matrix-relation items
        bool all_matrix = true;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const Item it = p.direct[first + g];
            if (it.row >= 0 && a.t[it.term].n_other != 1) all_matrix = false;
        }
#pragma unroll
        for (int g = 0; g < G; g++)
            staged[g] = (g < Geo<DP>::NSTAGE && (all_matrix || g == 0)) ? prestage<DP>(a, p.direct[first + g], lane, wl, g) : false;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const Item it = p.direct[first + g];
            rows[g] = it.row;
#pragma unroll
            for (int b = 0; b < NB; b++) accg[g][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < DB; I++) bredg[g][I] = 0.0;
            if (it.row >= 0 && it.count > 0)                                                            // wave-uniform
                accumulate_any<DP>(a, it, lane, wl, staged[g] ? g : 0, staged[g], accg[g], bredg[g]);
            STAMP(1 + g);
        }
        {
            int64_t r = -1;
#pragma unroll
            for (int g = 0; g < G; g++) r = (grp == g) ? rows[g] : r;
            if (r >= 0 && a.D - 1 - c >= 0 && !DUMP)
                znorm = bdf_normal(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)r, a.D - 1 - c);
        }
        {
            d4 lamc[NB];                                  // loaded after the gathers so that it is not live during them
            load_prior_c<DP>(a, lane, lamc);
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int b = 0; b < NB; b++) accg[g][b] += lamc[b];
        }
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
                // select form (not a divergent block): the column array is updated in place, no second copy of it
                const bool mine = (grp == g);
#pragma unroll
                for (int i = 0; i < DP; i++) {
                    const double v = wl[i * LD + c];
                    col[i] = (g == 0 || mine) ? v : col[i];
                }
                const double vb = wl[DP * LD + c];
                bj = mine ? vb : bj;
                myrow = mine ? rows[g] : myrow;
            }
        }
    }
    wave_sync();
    STAMP(3);
    {
        const int ec = a.D - 1 - c;
        if (myrow >= 0 && ec >= 0) bj += a.prior_b[(a.mu_is_matrix ? myrow * a.D : 0) + ec];
        else if (myrow < 0) {       // only the split-row finisher has unused lane groups
#pragma unroll
            for (int i = 0; i < DP; i++) col[i] = (i == c) ? 1.0 : 0.0;
            bj = 0.0;
        }
    }
    finish_rows<DP, DUMP>(a, myrow, col, bj, znorm, wl, lane, wid);
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
    int32_t *arrived_dev = nullptr;
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
        if (n_items <= 1) {
            Item it{row, 0, 0, 0, -1, -1, 0};
            for (int r = 0; r < key.n_terms; r++) {
                const auto &rp = rels[r]->idx[key.mode[r]].rowptr;
                const int64_t n = rp[(size_t)row + 1] - rp[(size_t)row];
                if (n > 0) { it.term = r; it.q_begin = rp[(size_t)row]; it.count = (int32_t)n; }
            }
            direct.push_back(it);
        } else {
            if (n_items == 0) {          // all-split mode, empty row: one empty item so that the row has a slot
                split.push_back(Item{row, 0, 0, 0, (int32_t)split.size(), (int32_t)srows.size(), 0});
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
                    split.push_back(Item{row, r, b0, (int32_t)(b1 - b0), (int32_t)split.size(), (int32_t)srows.size(), 0});
                }
            }
            srows.push_back(sr);
        }
    }
    // a wave finishes G rows at once: fill the last wave by repeating its last row (same result written twice)
    while (!direct.empty() && direct.size() % (size_t)G) direct.push_back(direct.back());
    int rc;
    if ((rc = to_device(direct, &plan.direct_dev)) || (rc = to_device(split, &plan.split_dev)) ||
        (rc = to_device(srows, &plan.rows_dev)))
        return rc;
    BDF_HIP(hipMalloc((void **)&plan.partials_dev, std::max<size_t>(split.size() * (size_t)psz * sizeof(double), 8)));
    BDF_HIP(hipMalloc((void **)&plan.arrived_dev, std::max<size_t>(srows.size() * sizeof(int32_t), 8)));
    BDF_HIP(hipMemset(plan.arrived_dev, 0, std::max<size_t>(srows.size() * sizeof(int32_t), 8)));
    plan.dev.direct = plan.direct_dev; plan.dev.n_direct = (int32_t)direct.size();
    plan.dev.split = plan.split_dev;   plan.dev.n_split = (int32_t)split.size();
    plan.dev.rows = plan.rows_dev;     plan.dev.n_split_rows = (int32_t)srows.size();
    plan.dev.partials = plan.partials_dev;
    plan.dev.arrived = plan.arrived_dev;
    return BDF_OK;
}

template <int DP>
int launch(bdf_ctx *ctx, const SampleArgs &a, const PlanDev &p, bool dump)
{
    constexpr int G = Geo<DP>::G, WPB = Geo<DP>::WPB;
    const int64_t waves = (int64_t)p.n_split + p.n_direct / G;
    if (waves > 0) {
        dim3 grid((unsigned)((waves + WPB - 1) / WPB)), block(64 * WPB);
        if (dump) hipLaunchKernelGGL((k_rows<DP, true>), grid, block, 0, ctx->stream, a, p);
        else      hipLaunchKernelGGL((k_rows<DP, false>), grid, block, 0, ctx->stream, a, p);
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
            (void)hipFree(kv->second.direct_dev); (void)hipFree(kv->second.split_dev); (void)hipFree(kv->second.rows_dev);
            (void)hipFree(kv->second.partials_dev); (void)hipFree(kv->second.arrived_dev);
            kv = plans.erase(kv);
        } else {
            ++kv;
        }
    }
    if (rel_serial == 0) g_caches.erase(it);
}

int bdf_launch_sample_rows(bdf_ctx *ctx, const SampleArgs &a_in, const bdf_rel *const *rels, const int *modes, int shard,
                           int n_shards, bool dump)
{
    SampleArgs a = a_in;
    {
        // prior part of b: Lambda mu (one vector) or Lambda mu_i for every row (per-row prior means, macau.jl:104)
        const int64_t N = rels[0]->dims[modes[0]];
        const int64_t nr = a.mu_is_matrix ? N : 1;
        void *pb;
        int rc = bdf_scratch(ctx, (size_t)nr * a.D * sizeof(double), &pb);
        if (rc) return rc;
        if (nr * a.D > 0) {
            hipLaunchKernelGGL(k_prior_b, dim3((unsigned)((nr * a.D + 3) / 4)), dim3(256), 0, ctx->stream, a.D, nr, a.Lambda,
                               a.mu, a.mu_is_matrix, (double *)pb);
            BDF_HIP(hipGetLastError());
        }
        a.prior_b = (const double *)pb;
    }
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
