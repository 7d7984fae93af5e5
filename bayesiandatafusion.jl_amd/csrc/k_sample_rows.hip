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
//   * a row's observations are cut into ITEMS of at most T observations; ONE WAVEFRONT PER ITEM.
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
// Finishing happens IN THE ACCUMULATOR LAYOUT, with the matrix never leaving the registers the MFMAs left it in:
// lane (j = l & 15, h = l >> 4), register r of block (I, J) holds element (16 I + h + 4 r, 16 J + j) -- a lane owns
// DB columns (j, 16 + j, ...) and of each the rows of its class h (mod 4): 12 doubles for D <= 32.  Step k of the
// right-looking factorisation needs, in lane (j, h), the entries of column k in the lane's rows -- they sit in lane
// (k % 16, h), same row of 16 lanes, same registers: a DPP row broadcast folded into the fma (v_fmac_f64_dpp
// row_newbcast) -- and the multipliers of the lane's columns, read from the copy of column k that its four owner
// lanes put in LDS (the packed factor that the backward solve reads anyway).  b rides along as one more matrix row,
// which makes the forward solve part of the factorisation.  The low register count (about a third of a
// column-per-lane layout) is what lets 5-6 waves share a SIMD and hide each other's dependent-step latencies.
#include "bdf_common.h"
#include "wave_linalg.h"
#include "c_layout_chol.h"
#include "dpp_rows16.h"
#include "dpp_rows32.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#ifndef BDF_CHOL_BLOCKED
#define BDF_CHOL_BLOCKED 1        // the row's factorisation in 16-column panels: multipliers without an LDS round trip, trailing update on the matrix cores (c_layout_chol.h); 0: the plain right-looking variant
#endif
#ifndef BDF_K1_KS
#define BDF_K1_KS 2               // k-steps (4 observations each) per pipelined trip, matrix relations
#endif
#ifndef BDF_K1_KS64
#define BDF_K1_KS64 2             // ... at D > 32
#endif

#ifdef BDF_K1_SPANS      // diagnostic build: per wave of every launch {start, end, wait for the prior} (s_memrealtime: the 100 MHz clock all XCDs share -- s_memtime is per XCD; plain stores)
#define SPAN_BEGIN() do { if (lane == 0 && a.b_dump && wid < 8192) ((unsigned long long *)a.b_dump)[wid * 3] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SPAN_END() do { if (lane == 0 && a.b_dump && wid < 8192) ((unsigned long long *)a.b_dump)[wid * 3 + 1] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SPAN_WAIT(t0) do { if (lane == 0 && a.b_dump && wid < 8192) ((unsigned long long *)a.b_dump)[wid * 3 + 2] = (unsigned long long)__builtin_amdgcn_s_memrealtime() - (t0); } while (0)
#else
#define SPAN_BEGIN() do { } while (0)
#define SPAN_END() do { } while (0)
#define SPAN_WAIT(t0) do { } while (0)
#endif
#ifdef BDF_K1_STAMPS
#define STAMP(slot) do { if (lane == 0 && a.b_dump && wid < 65536) ((unsigned long long *)a.b_dump)[wid * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

namespace {

#ifndef BDF_K1_LOCAL64
#define BDF_K1_LOCAL64 1          // D > 32: 8.6 KB of LDS per wave instead of 17.9 (c_layout_chol.h: GeoL)
#endif
template <int DP>
struct K1Local { static constexpr bool value = (DP == 64) && BDF_K1_LOCAL64 && BDF_CHOL_BLOCKED; };

struct Item {             // one wave's accumulation work
    int32_t row;          // entity row: where the sample is written (the row's position in the factor matrix)
    int32_t term;
    int64_t q_begin;      // first observation (index into the term's CSR arrays)
    int32_t count;        // observations in this item
    int32_t slot;         // partial slot, or -1 for a direct row
    int32_t srow;         // index of the row in the split-row table (split items)
    int32_t orig;         // the row's ORIGINAL id: keys its random stream (== row unless the relation was created with a layout)
};

struct SplitRow {
    int32_t row;
    int32_t slot_begin, n_slots;
    int32_t _pad;
};

struct PlanDev {
    const Item *direct;   int32_t n_direct;
    const Item *split;    int32_t n_split;
    const SplitRow *rows; int32_t n_split_rows;
    double *partials;                            // n_split * PSZ doubles
    int32_t *arrived;                            // per split row: items that have published their partial (self-resetting)
    const int32_t *order;                        // launch order: wave w takes item order[w] of [split | direct]
};


// ---- accumulate one item, register path (any D, per-observation baselines): acc (MFMA C layout, lower block-triangle)
// and bred (the item's part of b) ------------------------------------------------------------------------------------
// Software pipeline over "trips" of 4*KS observations: the other-mode ids and values of trip t+2 and the gathered
// factor rows of trip t+1 are in flight while the MFMAs of trip t issue.  Lane (j = l & 15, h = l >> 4) handles
// observations h, h+4, h+8, ... of the item and elements 16 I + j of their factor rows (index-reversed).
template <int DP, int NO>
__device__ __forceinline__ void accumulate_reg(const SampleArgs &a, const Item &it, int lane, d4 (&acc)[Geo<DP>::NB],
                                  double (&bred)[Geo<DP>::DB])
{
    constexpr int DB = Geo<DP>::DB, NB = Geo<DP>::NB;
    constexpr int KS = (NO == 1) ? BDF_K1_KS : (NO == 2 ? 2 : 1);   // k-steps (of 4 observations) per trip
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
    const double alpha = term_alpha(T);
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] *= alpha;
#pragma unroll
    for (int I = 0; I < DB; I++) {
        double v = bpart[I] * alpha;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        bred[I] = v;
    }
}

// ---- accumulate one item, lean register path: shared baseline (no per-observation linear_values), at most two other
// modes, factor matrices below 4 GiB with fewer than 2^24 rows (TermDev::lean, checked by the host) -----------------------
// Same pipeline as accumulate_reg, with what the general path pays per load taken out: wave-uniform (SGPR) bases with
// 32-bit byte offsets, row offsets by one 24-bit mad, no predicated loads (indices are clamped to the item instead, and
// only the item's last trip masks its operands).
// CODED (a launch with one two-mode relation whose values are at most BDF_K1_CODES distinct numbers -- ratings): the other-mode
// id and the value's 8-bit code arrive as ONE 32-bit word per observation (TermDev::packed), the value minus the mean comes
// from the wave's own table in LDS (the packed factor's space, idle until the factorisation) when it is used.  No value is held in registers two trips ahead: 70 instead of 80 VGPRs, SEVEN
// resident waves per SIMD instead of six, and a third fewer memory instructions per trip.  Same arithmetic, same results.
template <int DP, int NO, bool FULL, bool WIDE = false, bool CODED = false>
__device__ __forceinline__ void accumulate_lean(const SampleArgs &a, const Item &it, int lane, d4 (&acc)[Geo<DP>::NB],
                                       double (&bred)[Geo<DP>::DB], const double *tab = nullptr)
{
    static_assert(!CODED || (NO == 1 && !WIDE), "coded values: one two-mode relation, 32-bit row offsets");
    constexpr int DB = Geo<DP>::DB, NB = Geo<DP>::NB;
    constexpr int KS = (NO == 1) ? (DP == 64 ? BDF_K1_KS64 : BDF_K1_KS) : 1;
    const TermDev &T = a.t[it.term];
    const int D = FULL ? DP : a.D;
    const int j = lane & 15, h = lane >> 4;
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    double bpart[DB];
    uint32_t eoff[DB];                        // byte offset in a factor row of reversed element 16 I + j
    bool eok[DB];
#pragma unroll
    for (int I = 0; I < DB; I++) {
        bpart[I] = 0.0;
        const int ec = D - 1 - (16 * I + j);
        eok[I] = FULL || ec >= 0;
        eoff[I] = (uint32_t)(ec >= 0 ? ec : 0) * 8u;
    }
    const uint32_t n = (uint32_t)it.count, rowb = (uint32_t)D * 8u;
    const uint32_t ntrips = (n + 4 * KS - 1) / (4 * KS);
    const char *ids[NO], *fac[NO];
#pragma unroll
    for (int m = 0; m < NO; m++) {
        ids[m] = CODED ? (const char *)(T.packed + it.q_begin) : (const char *)(T.colidx + (int64_t)m * T.nnz + it.q_begin);
        fac[m] = (const char *)T.fac[m];
    }
    const char *vals = (const char *)(T.vals + it.q_begin);
    const double mean = T.mean;
    double tab_v = 0.0;
    if (CODED && lane < BDF_K1_CODES) tab_v = T.table[lane] - mean;      // this wave's copy of the table: value - mean by code

    // two register sets, used alternately by even and odd trips (no rotation copies: a copy would have to wait for
    // the load it moves).  Trip t multiplies set t%2; the gathers of trip t+1 fill the other set; the ids and values of
    // trip t+2 are loaded into set t%2 once trip t has used it.
    uint32_t ix[2][KS][NO];
    double rr[2][KS];
    double w[2][KS][NO][DB];
// observation of (trip t, k-step k, lane group h): t * 4 KS + KS h + k -- a lane group's KS observations of a trip are
// neighbours, so that the coded variant fetches their words with ONE load (the packed array carries a spare word at its end)
#define OBS(t, k) ((t) * (4 * KS) + KS * h + (k))
#define LOAD_IDX(t, S)                                                                          \
    if (CODED && KS == 2) {                                                                     \
        uint32_t o = OBS(t, 0);                                                                 \
        o = (o < n ? o : n - 1) * 4u;                                                           \
        const uint2 pw = *(const uint2 *)(ids[0] + o);                                          \
        ix[S][0][0] = pw.x; ix[S][KS - 1][0] = pw.y;                                            \
    } else if (KS == 2 && NO == 1) {          /* the pair's ids and values as one load each (the arrays carry a spare entry) */ \
        uint32_t o = OBS(t, 0);                                                                 \
        o = (o < n ? o : n - 1) * 4u;                                                           \
        const uint2 pi = *(const uint2 *)(ids[0] + o);                                          \
        const d2 pv = *(const d2 *)(vals + 2u * o);                                             \
        ix[S][0][0] = pi.x; ix[S][KS - 1][0] = pi.y;                                            \
        rr[S][0] = pv[0]; rr[S][KS - 1] = pv[1];                                                \
    } else {                                                                                    \
    _Pragma("unroll") for (int k = 0; k < KS; k++) {                                            \
        uint32_t o = OBS(t, k);                                                                 \
        o = (o < n ? o : n - 1) * 4u;                                                           \
        _Pragma("unroll") for (int m = 0; m < NO; m++) ix[S][k][m] = *(const uint32_t *)(ids[m] + o); \
        if (!CODED) rr[S][k] = *(const double *)(vals + 2u * o);                                \
    }                                                                                           \
    }
#define LOAD_DATA(S)                                                                            \
    _Pragma("unroll") for (int k = 0; k < KS; k++)                                              \
        _Pragma("unroll") for (int m = 0; m < NO; m++)                                          \
            _Pragma("unroll") for (int I = 0; I < DB; I++)                                      \
                w[S][k][m][I] = WIDE ? *(const double *)(fac[m] + ((uint64_t)ix[S][k][m] * rowb + eoff[I]))            \
                                     : *(const double *)(fac[m] + (__umul24(ix[S][k][m], rowb) + eoff[I]));
#define TRIP(t, C, X)                                                                           \
    {                                                                                           \
        LOAD_DATA(X)  /* unconditional (ids are clamped to the item): a branch here would cost exact waitcnts */ \
        double w_c[KS][DB];                                                                     \
        _Pragma("unroll") for (int k = 0; k < KS; k++)                                          \
            _Pragma("unroll") for (int I = 0; I < DB; I++) {                                    \
                double v = w[C][k][0][I];                                                       \
                _Pragma("unroll") for (int m = 1; m < NO; m++) v *= w[C][k][m][I];              \
                w_c[k][I] = v;                                                                  \
            }                                                                                   \
        if ((t) + 1 >= ntrips || !FULL) {     /* ragged last trip; padded elements when D < DP */ \
            _Pragma("unroll") for (int k = 0; k < KS; k++) {                                    \
                const bool valid = OBS(t, k) < n;                                               \
                _Pragma("unroll") for (int I = 0; I < DB; I++) w_c[k][I] = (valid && eok[I]) ? w_c[k][I] : 0.0; \
            }                                                                                   \
        }                                                                                       \
        _Pragma("unroll") for (int k = 0; k < KS; k++) {                                        \
            /* (__umul24 in LOAD_DATA takes the low 24 bits of the packed word: the id) */      \
            const double r = CODED ? tab[ix[C][k][0] >> 24] : rr[C][k] - mean;                  \
            int b = 0;                                                                          \
            _Pragma("unroll") for (int I = 0; I < DB; I++) {                                    \
                _Pragma("unroll") for (int J = 0; J <= I; J++) {                                \
                    acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(w_c[k][I], w_c[k][J], acc[b], 0, 0, 0);     \
                    b++;                                                                        \
                }                                                                               \
                bpart[I] = fma(w_c[k][I], r, bpart[I]);                                         \
            }                                                                                   \
        }                                                                                       \
        LOAD_IDX((t) + 2, C)                                                                    \
    }

    LOAD_IDX(0u, 0)
    LOAD_IDX(1u, 1)
    LOAD_DATA(0)
    // (its load was issued before the ids': it has arrived with them; a wave's LDS operations execute in order, so the reads
    // below need no wait for this write, and the factorisation's writes none for those reads)
    if (CODED && lane < BDF_K1_CODES) const_cast<double *>(tab)[lane] = tab_v;
    // trips go in pairs in one straight-line block (a branch between them lets the compiler sink the run-ahead loads to
    // their use); for an odd count the last one is empty: its operands are masked to zero
    for (uint32_t t = 0; t < ntrips; t += 2) {
        TRIP(t, 0, 1)
        TRIP(t + 1, 1, 0)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the run-ahead loads of the last trips
#undef LOAD_IDX
#undef LOAD_DATA
#undef TRIP
#undef OBS
    const double alpha = term_alpha(T);
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] *= alpha;
#pragma unroll
    for (int I = 0; I < DB; I++) {
        double v = bpart[I] * alpha;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        bred[I] = v;
    }
}

// path and other-mode count are wave-uniform.  MATRIX: the kernel variant for launches whose terms are all two-mode
// relations on the lean path -- without the tensor and general gathers the D <= 32 kernel needs 78 registers instead of
// 92 (6 resident waves per SIMD instead of 5, and room beside 5 of them for a wave of the prediction update)
template <int DP, bool MATRIX, bool CODED = false>
__device__ __forceinline__ void accumulate_any(const SampleArgs &a, const Item &it, int lane, d4 (&acc)[Geo<DP>::NB],
                                      double (&bred)[Geo<DP>::DB], const double *tab = nullptr)
{
    if constexpr (CODED) {                   // one two-mode relation, lean gather, coded values (checked by the host)
        if (a.D == DP) accumulate_lean<DP, 1, true, false, true>(a, it, lane, acc, bred, tab);
        else accumulate_lean<DP, 1, false, false, true>(a, it, lane, acc, bred, tab);
        return;
    }
    const int no = a.t[it.term].n_other;
    if constexpr (DP == 64) {
        if (a.t[it.term].lean == 2) {        // a factor matrix of 4 GiB or more (e.g. 10M rows at D = 64): 64-bit row offsets
            if (a.D == DP) {
                if (MATRIX || no == 1) accumulate_lean<DP, 1, true, true>(a, it, lane, acc, bred);
                else accumulate_lean<DP, 2, true, true>(a, it, lane, acc, bred);
            } else {
                if (MATRIX || no == 1) accumulate_lean<DP, 1, false, true>(a, it, lane, acc, bred);
                else accumulate_lean<DP, 2, false, true>(a, it, lane, acc, bred);
            }
            return;
        }
    }
    if constexpr (MATRIX) {                  // every term of the launch: two modes, lean gather (checked by the host)
        if (a.D == DP) accumulate_lean<DP, 1, true>(a, it, lane, acc, bred);
        else accumulate_lean<DP, 1, false>(a, it, lane, acc, bred);
        return;
    }
    if (a.t[it.term].lean == 1) {
        if (a.D == DP) {
            if (no == 1) accumulate_lean<DP, 1, true>(a, it, lane, acc, bred);
            else accumulate_lean<DP, 2, true>(a, it, lane, acc, bred);
        } else {
            if (no == 1) accumulate_lean<DP, 1, false>(a, it, lane, acc, bred);
            else accumulate_lean<DP, 2, false>(a, it, lane, acc, bred);
        }
        return;
    }
    if (no == 1) accumulate_reg<DP, 1>(a, it, lane, acc, bred);
    else if (no == 2) accumulate_reg<DP, 2>(a, it, lane, acc, bred);
    else accumulate_reg<DP, 3>(a, it, lane, acc, bred);
}

// ---- prior: a small pre-launch writes Lambda mu_i (the prior part of b) and the image of the index-reversed Lambda in
// the accumulator layout ([block * 4 + r][lane], identity on the padding), which every wave adds with coalesced loads. ----
__global__ __launch_bounds__(256) void k_prior(int D, int DP, int64_t nrows, const double *Lambda, const double *mu,
                                               int mu_is_matrix, double *out_b, double *out_c)
{
    // groups of eight lanes 0 .. nrows*D-1: out_b[row*D + e] = sum_i Lambda[e][i] mu_row[i]  (nrows = 1 for a shared prior
    // mean); lane part p adds i = p, p+8, ... in order, then a three-step butterfly (the order k_hyper_sample uses too).
    // Then NB*4 waves, one per (block, register) of the image.
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int DB = DP / 16;
    const int64_t nb8 = (nrows * D + 7) / 8;                  // waves used by the first part
    if ((t >> 6) < nb8) {
        const int64_t o = t >> 3;
        const int part = (int)(t & 7);
        double v = 0.0;
        if (o < nrows * D) {
            const int64_t row = o / D;
            const int e = (int)(o % D);
            const double *m = mu_is_matrix ? mu + row * D : mu;
            for (int i = part; i < D; i += 8) v = fma(Lambda[e + (int64_t)i * D], m[i], v);
        }
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (part == 0 && o < nrows * D) out_b[o] = v;
        return;
    }
    const int64_t idx = (t >> 6) - nb8 + nrows * D;
    const int e = (int)(idx - nrows * D);
    if (e >= DB * (DB + 1) / 2 * 4) return;
    const int b = e >> 2, r = e & 3;
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= b) I++;
    const int J = b - I * (I + 1) / 2;
    const int row = 16 * I + (lane >> 4) + 4 * r, colm = 16 * J + (lane & 15);
    const int er = D - 1 - row, ecm = D - 1 - colm;
    double v = (row == colm) ? 1.0 : 0.0;
    if (er >= 0 && ecm >= 0) v = Lambda[er + (int64_t)ecm * D];
    else if (er >= 0 || ecm >= 0) v = 0.0;
    out_c[e * 64 + lane] = v;
}

// the sums of blocks b - 4 .. b are made before any later load is issued (a compiler fence that also pins the sums: at DP = 64 the
// 40 or 44 loads of a prior image / a partial slot all in flight beside the 80-register matrix are the kernel's register peak)
template <int NB>
__device__ __forceinline__ void batch_fence(d4 (&acc)[NB], int b)
{
#pragma unroll
    for (int q = 0; q < NB; q++)
        if (q <= b && q + 5 > b) {
#pragma unroll
            for (int r = 0; r < 4; r++) { double t = acc[q][r]; asm volatile("" : "+v"(t) : : "memory"); acc[q][r] = t; }
        }
}

// ---- sum the partials of a split row in slot order (fixed order: the result does not depend on which wave does it) ---
template <int DP>
__device__ __forceinline__ void sum_partials(const PlanDev &p, const SplitRow &sr, int lane, d4 (&acc)[Geo<DP>::NB],
                                    double (&bred)[Geo<DP>::DB])
{
    constexpr int DB = Geo<DP>::DB, NB = Geo<DP>::NB, PSZ = Geo<DP>::PSZ;
    constexpr int U = 1;                              // slots loaded per trip
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < DB; I++) bred[I] = 0.0;
    if constexpr (DP == 64 && BDF_K1_WAVES64 >= 3) {
        // a slot in two halves of 22 doubles (44 at once plus the 80-register matrix would leave nothing of a 168-register budget);
        // the same additions in the same order
        for (int s = 0; s < sr.n_slots; s++) {
            const double *src = p.partials + (int64_t)(sr.slot_begin + s) * PSZ;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                double v[NB * 2 + DB / 2];
#pragma unroll
                for (int e = 0; e < NB * 2; e++) v[e] = src[(half * NB * 2 + e) * 64 + lane];
#pragma unroll
                for (int I = 0; I < DB / 2; I++) v[NB * 2 + I] = src[NB * 4 * 64 + (half * (DB / 2) + I) * 16 + (lane & 15)];
#pragma unroll
                for (int e = 0; e < NB * 2; e++) acc[(half * NB * 2 + e) >> 2][(half * NB * 2 + e) & 3] += v[e];
#pragma unroll
                for (int I = 0; I < DB / 2; I++) bred[half * (DB / 2) + I] += v[NB * 2 + I];
                batch_fence<NB>(acc, half * (NB / 2) + NB / 2 - 1);
            }
        }
        return;
    }
    for (int s0 = 0; s0 < sr.n_slots; s0 += U) {
        double v[U][NB * 4 + DB];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = (s0 + u < sr.n_slots) ? s0 + u : sr.n_slots - 1;
            const double *src = p.partials + (int64_t)(sr.slot_begin + s) * PSZ;
#pragma unroll
            for (int e = 0; e < NB * 4; e++) v[u][e] = src[e * 64 + lane];
#pragma unroll
            for (int I = 0; I < DB; I++) v[u][NB * 4 + I] = src[NB * 4 * 64 + I * 16 + (lane & 15)];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
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
// independent: cdna_hip_programming.md Guideline 16).  The remaining waves take one direct row each. -----------------------
// One work item (index wi in [split items | direct items]) on one wave.
template <int DP, bool DUMP, bool MATRIX, bool CODED = false>
__device__ __forceinline__ void process_item(const SampleArgs &a, const PlanDev &p, const int64_t wid, const int lane, double *tri)
{
    double *const tab = tri;          // CODED: the wave's value table (BDF_K1_CODES doubles) sits in the packed factor's space until the factorisation
    using GG = Geo<DP>;
    constexpr int DB = GG::DB, NB = GG::NB, PSZ = GG::PSZ;
    const int j = lane & 15, h = lane >> 4;
    const int D = a.D;
    d4 acc[NB];
    double bv[DB];
    int64_t row;
    STAMP(0);
    SPAN_BEGIN();
#ifdef BDF_K1_STAMPS
    if (lane == 0 && a.b_dump && wid < 65536) {           // where the wave runs: HW_ID (wave, SIMD, CU, SH, SE) and XCC_ID
        ((unsigned long long *)a.b_dump)[wid * 16 + 9] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) |
            ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4) );
        ((unsigned long long *)a.b_dump)[wid * 16 + 10] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    }
#endif

    const bool is_split = wid < p.n_split;                      // wave-uniform
    const Item it = is_split ? p.split[wid] : p.direct[wid - p.n_split];
    row = it.row;
    // a direct row's normals (lane c < D draws number D-1-c of the row's stream) are drawn BEFORE its gathers: the
    // Philox / Box-Muller arithmetic then runs under the matrix-pipe-bound accumulation instead of after it
    double z = 0.0;
    const bool early_z = !DUMP && !is_split;
    if (early_z && lane < D) z = bdf_normal(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, D - 1 - lane);
    if (it.count > 0) accumulate_any<DP, MATRIX, CODED>(a, it, lane, acc, bv, tab);
    else {
#pragma unroll
        for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int I = 0; I < DB; I++) bv[I] = 0.0;
    }
    STAMP(1);
    if (is_split) {
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
                __hip_atomic_store(dst + NB * 4 * 64 + I * 16 + lane, bv[I], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // publish: every lane's write-through stores have completed, then one arrival
        const SplitRow sr = p.rows[it.srow];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(p.arrived + it.srow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != sr.n_slots - 1) { SPAN_END(); return; }       // not the last item of the row
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) p.arrived[it.srow] = 0;                  // ready for the next launch
        // the finisher's normals before the partial sums are loaded: the Box-Muller arithmetic needs ~40 registers
        if (!DUMP && lane < D) z = bdf_normal(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, D - 1 - lane);
        sum_partials<DP>(p, sr, lane, acc, bv);
        STAMP(2);
    }
    if (a.ready) {
        // launched without waiting for the hyperprior draw (bdf_gibbs_sweep: the draw runs on CUs this kernel never uses, so
        // it cannot be starved): poll its flag here, where the prior is first needed -- the gathers above have hidden most of
        // the wait -- and read the pack with agent-scope loads (past the non-coherent L2 lines of the previous sweep's pack)
        int spins = 0;
#ifdef BDF_K1_SPANS
        const unsigned long long t_poll = __builtin_amdgcn_s_memrealtime();
#endif
        while ((int32_t)(__hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ready_want) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 22)) { if (lane == 0) atomicOr_system(a.flag, 16); break; }      // bounded: ~seconds
        }
        SPAN_WAIT(t_poll);
        // (DP = 64: the image's 40 loads in batches of 20, so that no more than 40 registers of it are in flight beside the matrix)
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                acc[b][r] += __hip_atomic_load(a.prior_c + (b * 4 + r) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (DP == 64 && BDF_K1_WAVES64 >= 3 && b % 5 == 4) batch_fence<NB>(acc, b);
        }
#pragma unroll
        for (int J = 0; J < DB; J++) {
            const int ec = D - 1 - (16 * J + j);
            if (ec >= 0) bv[J] += __hip_atomic_load(a.prior_b + ec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int r = 0; r < 4; r++) acc[b][r] += a.prior_c[(b * 4 + r) * 64 + lane];
            if (DP == 64 && BDF_K1_WAVES64 >= 3 && b % 5 == 4) batch_fence<NB>(acc, b);
        }
#pragma unroll
        for (int J = 0; J < DB; J++) {
            const int ec = D - 1 - (16 * J + j);
            if (ec >= 0) bv[J] += a.prior_b[(a.mu_is_matrix ? row * D : 0) + ec];
        }
    }
    STAMP(3);

    if (DUMP) {
        // P~ and b of the row (bdf_row_system): element (i, c) of the reversed system is entry (D-1-i, D-1-c) of P
        int b = 0;
#pragma unroll
        for (int I = 0; I < DB; I++)
#pragma unroll
            for (int J = 0; J <= I; J++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ei = D - 1 - (16 * I + h + 4 * r), ec = D - 1 - (16 * J + j);
                    if (ei >= 0 && ec >= 0) {
                        a.P_dump[(row * D + ec) * D + ei] = acc[b][r];
                        if (I != J) a.P_dump[(row * D + ei) * D + ec] = acc[b][r];
                    }
                }
                b++;
            }
        if (h == 0) {
#pragma unroll
            for (int J = 0; J < DB; J++) {
                const int ec = D - 1 - (16 * J + j);
                if (ec >= 0) a.b_dump[row * D + ec] = bv[J];
            }
        }
        return;
    }

    STAMP(4);

    double A[NB * 4];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) A[b * 4 + r] = acc[b][r];
    if constexpr (DP == 64 && BDF_K1_WAVES64 >= 3) {
        // a boundary for the register allocator: the values that live through the factorisation start new live ranges here, so that
        // what the load phases above may have to keep in scratch under a three-wave budget is in registers again for the steps
#pragma unroll
        for (int b = 0; b < NB * 4; b++) asm volatile("" : "+v"(A[b]));
#pragma unroll
        for (int J = 0; J < DB; J++) asm volatile("" : "+v"(bv[J]));
        asm volatile("" : "+v"(z));
    }
    double ts[DB];                                // ts[J] in lane j: t_(16 J + j) once its step has passed; the last column's
#pragma unroll                                    // (and any column's before its step) is still in bv
    for (int J = 0; J < DB; J++) ts[J] = 0.0;
    constexpr bool LOCAL = K1Local<DP>::value;    // DP = 64: one panel of the factor in LDS at a time, the backward solve fed from the registers
    if (D < DP && !LOCAL) zero_packed_factor<DP>(tri, lane);
    if constexpr (LOCAL)
        factor_all_blocked_local<DP>(A, bv, ts, tri, j, h, D, std::make_integer_sequence<int, DP - 1>{});
    else if constexpr (BDF_CHOL_BLOCKED)
        factor_all_blocked<DP>(A, bv, ts, tri, j, h, D, std::make_integer_sequence<int, DP - 1>{});
    else
        factor_all<DP>(A, bv, ts, tri, j, h, D, std::make_integer_sequence<int, DP - 1>{});
    STAMP(5);

    // lane c = column c: pivot d_c from the packed factor, t_c (the forward solve, unscaled) from the extra row
    const int cK = (lane < DP) ? (lane >> 4) : 0;
    const typename GG::ColRT cr = GG::col_rt(lane < DP ? lane : 0);       // this lane's column of the packed factor
    wave_sync();
    double dv = 1.0, tv = 0.0;
    if (lane < D) dv = LOCAL ? tri[GeoL<DP>::PIV + lane] : tri[cr.cbase + (lane & 3) * cr.nr4];      // the diagonal entry is the first of its row class
    if (!(dv > 0.0)) atomicOr_system(a.flag, 1);                      // a pivot that is not positive (or NaN): not positive definite
#pragma unroll
    for (int J = 0; J < DB; J++) tv = (lane < D && cK == J) ? ts[J] : tv;
    const double rdv = fast_rcp(dv);
    // L w = b, y = w + z carried as yh = y sqrt(d) = t + z sqrt(d);  then Lt' x = yh
    double yh = fma(z, dv * fast_rsqrt(dv), tv);
    if constexpr (LOCAL) {
        backward_rows<DP>(A, yh, rdv, tri, lane);
    } else {
        unsigned colq[4];                          // LDS byte addresses: row i of this lane's column at colq[i & 3] + 8 (i >> 2)
#pragma unroll
        for (int q = 0; q < 4; q++)
            colq[q] = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(tri + cr.cbase + q * cr.nr4 - cr.q);
        backward_all<DP>(yh, rdv, colq, std::make_integer_sequence<int, DP / 16>{});
    }
    if (lane < D) a.out[row * D + (D - 1 - lane)] = yh * rdv;
    STAMP(8);
    SPAN_END();
}

template <int DP, bool DUMP, bool MATRIX, bool CODED = false>
__global__ __launch_bounds__(64 * Geo<DP>::WPB, CODED ? Geo<DP>::WAVES_CODED : (MATRIX ? Geo<DP>::WAVES_MATRIX : Geo<DP>::WAVES))
void k_rows(SampleArgs a, PlanDev p)
{
    using GG = Geo<DP>;
    constexpr int WPB = GG::WPB;
    constexpr int WLDS = K1Local<DP>::value ? GeoL<DP>::WAVE_LDS : GG::WAVE_LDS;
    __shared__ __attribute__((aligned(16))) double lds[WPB * WLDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w = (int64_t)blockIdx.x * WPB + wave;
    if (w < (int64_t)p.n_split + p.n_direct)
        process_item<DP, DUMP, MATRIX, CODED>(a, p, p.order[w], lane, lds + wave * WLDS);
}

// ---- D <= 16, short rows of ONE two-mode relation: FOUR ROWS PER WAVE -------------------------------------------------
// At D <= 16 a row of ten observations costs the wave-per-row kernel ~570 vector and ~340 scalar instructions, nearly all of
// them per-row overhead that 64 lanes execute for one 16 x 16 system (normals, index arithmetic, 15 factorisation steps on
// a quarter-filled block): that kernel is issue-bound there (the reference's own benchmark shape: 1.5 M rows of ~10
// observations).  Here every 16-lane row of the wave owns one entity row; lane j of it holds COLUMN j of the index-reversed
// system (16 doubles) and b_j.  Observations come 16 at a time (lane j loads the id and value of observation c0 + j, the ids
// are broadcast inside the 16-lane row by DPP and the 16 gathers are all in flight); the rank-1 updates, the LDL'
// factorisation with the forward solve riding along as one more row, and the backward solve are DPP row-broadcast fmas
// (v_fmac_f64_dpp row_newbcast: lane k of each 16-lane row).  Same arithmetic contract as k_rows: the sample is
// x~ = L~^-T (D^-1 L~^-1 b~ + D^-1/2 z~) of the reversed system P~ = L~ D L~', lane j drawing number D - 1 - j of the row's
// stream; sums over observations run in observation order.
struct SmallItem {
    int32_t row;          // where the sample is written; -1: no row (padding of the last wave)
    int32_t orig;         // the row's original id (random stream)
    int64_t q_begin;
    int32_t count, _pad;
};

// eight observations of a chunk: ids broadcast inside the 16-lane row, all eight gathers issued (observations past the row's
// end gather row 0 and are masked to zero), then the rank-1 updates
template <int H, int K>
__device__ __forceinline__ void small_gather(double (&v)[8], uint32_t idw, const char *fac, uint32_t rowb, uint32_t eoff)
{
    if constexpr (K < 8) {
        v[K] = *(const double *)(fac + (__umul24(row_bcast_u32<8 * H + K>(idw), rowb) + eoff));      // (lean gather: 32-bit offsets)
        small_gather<H, K + 1>(v, idw, fac, rowb, eoff);
    }
}
template <int DR, int H, int K>
__device__ __forceinline__ void small_chunk(double (&A)[16], double &b, const double (&v)[8], double r, int n_here, bool jok)
{
    if constexpr (K < 8) {
        const double vk = (jok && 8 * H + K < n_here) ? v[K] : 0.0;
        b = fma(vk, row_bcast_f64<8 * H + K>(r), b);
        small_rank1<DR, 0>(A, vk);
        small_chunk<DR, H, K + 1>(A, b, v, r, n_here, jok);
    }
}

#ifndef BDF_SMALL_BLOCKS
#define BDF_SMALL_BLOCKS 1
#endif
template <bool CODED, int DR>
__global__ __launch_bounds__(256, BDF_SMALL_BLOCKS) void k_rows_small(SampleArgs a, const SmallItem *items, int64_t n_items)
{
    const int lane = threadIdx.x & 63, j = lane & 15;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w * 4 >= n_items) return;
    const SmallItem it = items[w * 4 + (lane >> 4)];
    const bool live = it.row >= 0;
    const int D = a.D;
    const TermDev &T = a.t[0];
    const int ec = D - 1 - j;                   // natural index of reversed element j
    const bool jok = ec >= 0;
    double z = 0.0;
    if (live && jok) z = bdf_normal(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, ec);
    double A[16];
#pragma unroll
    for (int i = 0; i < 16; i++) A[i] = 0.0;
    double b = 0.0;
    const int n = live ? it.count : 0;
    int nmax = n;
    nmax = max(nmax, __shfl_xor(nmax, 16));
    nmax = max(nmax, __shfl_xor(nmax, 32));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    const char *fac = (const char *)T.fac[0];
    const uint32_t rowb = (uint32_t)D * 8u, eoff = (uint32_t)(jok ? ec : 0) * 8u;
    const double mean = T.mean;
    for (int c0 = 0; c0 < nmax; c0 += 16) {
        const int o = c0 + j;
        uint32_t idw = 0;
        double r = 0.0;
        if (o < n) {
            if (CODED) {
                const uint32_t pw = T.packed[it.q_begin + o];
                idw = pw & 0xffffffu;
                r = T.table[pw >> 24] - mean;
            } else {
                idw = (uint32_t)T.colidx[it.q_begin + o];
                r = T.vals[it.q_begin + o] - mean;
            }
        }
        const int left = nmax - c0;                       // (wave-uniform: the longest of the four rows)
        double v0[8];
        small_gather<0, 0>(v0, idw, fac, rowb, eoff);
        if (left > 8) {
            double v1[8];
            small_gather<1, 0>(v1, idw, fac, rowb, eoff);
            small_chunk<DR, 0, 0>(A, b, v0, r, n - c0, jok);
            small_chunk<DR, 1, 0>(A, b, v1, r, n - c0, jok);
        } else small_chunk<DR, 0, 0>(A, b, v0, r, n - c0, jok);
    }
    // prior: the image of the index-reversed Lambda is in k_rows' accumulator layout -- element (i, j) of a one-block system
    // sits at [(i / 4) * 64 + (i % 4) * 16 + j]; read past the caches when the draw was polled for (as k_rows does)
    const double alpha = term_alpha(T);
    if (a.ready) {
        int spins = 0;
        while ((int32_t)(__hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ready_want) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 22)) { if (lane == 0) atomicOr_system(a.flag, 16); break; }
        }
#pragma unroll
        for (int i = 0; i < DR; i++)
            A[i] = fma(alpha, A[i], __hip_atomic_load(a.prior_c + (i / 4) * 64 + (i % 4) * 16 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        b = fma(alpha, b, jok ? __hip_atomic_load(a.prior_b + ec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0);
    } else {
#pragma unroll
        for (int i = 0; i < DR; i++) A[i] = fma(alpha, A[i], a.prior_c[(i / 4) * 64 + (i % 4) * 16 + j]);
        b = fma(alpha, b, (jok && live) ? a.prior_b[(a.mu_is_matrix ? (int64_t)it.row * D : 0) + ec] : 0.0);
    }
#pragma unroll
    for (int i = 0; i < DR; i++)
        if (i >= D || !jok) A[i] = (i == j) ? 1.0 : 0.0;          // padding: identity
    if (!jok) b = 0.0;
    double dj = 1.0;
    small_factor<DR, 0>(A, b, dj, j);
    if (live && jok && !(dj > 0.0)) atomicOr_system(a.flag, 1);
    const double rdj = fast_rcp(dj);
    double y = fma(z, fast_rsqrt(dj), b * rdj);
    small_backward<DR - 1>(A, y, rdj, j);
    if (live && jok) a.out[(int64_t)it.row * D + ec] = y;
}

// ---- host: the plan (items, split rows, slab) for a (terms, row list) combination, cached per context ---------------
struct PlanKey {
    uint64_t rel[BDF_MAX_TERMS];      // relation serials
    int mode[BDF_MAX_TERMS];
    int n_terms, DP, T, Tp;
    int shard, n_shards;
    int small;                        // > 0: rows of at most this many observations go to k_rows_small (four rows per wave)
    int lr;                           // > 0: rows of at most this many observations go to k_rows_lr (the low-rank sampler, k_rows_lr.hip)
    int lr32;                         // > lr: rows of lr + 1 .. lr32 observations too (k_rows_lr32: two observations per lane, D > 32)
    int64_t lr_min, lr_other;         // ... if the launch has at least lr_min of them, and at least half as many as the opposite entity has rows
    int col;                          // > 0: the rows of k_rows go to k_rows_col instead (four rows per wave, column layout), cut into pieces of at most this size
    int col_slots;                    // ... dealt to at most this many waves
    bool operator<(const PlanKey &o) const { return memcmp(this, &o, sizeof(PlanKey)) < 0; }
};

struct Plan {
    PlanDev dev;
    SmallItem *small_dev = nullptr;
    int64_t n_small = 0;              // entries of small_dev (a multiple of 4)
    SmallItem *lr_dev = nullptr;      // the rows of the low-rank sampler (same record), and their positions for the back-transform
    int32_t *lr_rows_dev = nullptr;
    int64_t n_lr = 0, n_lr_padded = 0;   // rows of the low-rank sampler in all; records of the rows of at most key.lr observations (a multiple of 4)
    int64_t n_lr32_padded = 0;           // ... and of the rows of key.lr + 1 .. key.lr32 observations, behind them in lr_dev
    Item *direct_dev = nullptr, *split_dev = nullptr;
    SplitRow *rows_dev = nullptr;
    int32_t *order_dev = nullptr;
    double *partials_dev = nullptr;
    int32_t *arrived_dev = nullptr;
    bdf_col_plan col;                 // the rows of k_rows_col (K1c)
    int64_t rows_lr = 0, rows_small = 0, rows_col = 0, rows_k1 = 0;      // how the plan's rows are shared out (bdf_ctx_rows_dispatch)
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

constexpr int64_t MAX_PIECES = 64;

// one row of the launch as the plan sees it: where its sample goes, its original id, and per term its observations in
// that term's device arrays
struct RowRef {
    int32_t out, orig;
    int64_t qb[BDF_MAX_TERMS];
    int64_t cnt[BDF_MAX_TERMS];
};

int build_plan(bdf_ctx *ctx, const PlanKey &key, const std::vector<RowRef> &rows, int psz, bool lr_on, Plan &plan)
{
    const int T = key.T;
    std::vector<Item> direct, split;
    std::vector<SmallItem> small, lr, lr32;
    std::vector<SplitRow> srows;
    auto row_total = [&](const RowRef &rr) { int64_t n = 0; for (int r = 0; r < key.n_terms; r++) n += rr.cnt[r]; return n; };
    // a row that neither k_rows_small nor the low-rank sampler takes
    auto row_is_k1 = [&](const RowRef &rr) {
        int nz = 0;
        for (int r = 0; r < key.n_terms; r++) nz += rr.cnt[r] > 0;
        const int64_t n = row_total(rr);
        if (nz <= 1 && key.small > 0 && n <= key.small) return false;
        if (nz <= 1 && lr_on && n <= std::max(key.lr, key.lr32)) return false;
        return true;
    };
    std::vector<bdf_row_ref> crows;
    for (const RowRef &rr : rows) {
        if (key.col > 0 && row_is_k1(rr)) { crows.push_back(bdf_row_ref{rr.out, rr.orig, rr.qb[0], rr.cnt[0]}); continue; }
        const int32_t row = rr.out;
        int n_items = 0;
        for (int r = 0; r < key.n_terms; r++) n_items += (int)std::min<int64_t>((rr.cnt[r] + T - 1) / T, MAX_PIECES);
        // (at most MAX_PIECES per relation: the row's finisher adds the partial sums one slot after the other, ~0.5 us each
        // -- a 78,000-observation row of config C5 in 128-observation pieces would keep it busy for 0.3 ms)
        // a row that is split anyway is cut into smaller pieces than the longest whole row: the launch ends with the split
        // rows (their pieces gather at a sixth of the matrix pipe each, then one wave sums and finishes the row)
        const int Tp = n_items > 1 ? key.Tp : T;
        if (n_items > 1 && Tp != T) {
            n_items = 0;
            for (int r = 0; r < key.n_terms; r++) n_items += (int)std::min<int64_t>((rr.cnt[r] + Tp - 1) / Tp, MAX_PIECES);
        }
        if (n_items <= 1) {
            Item it{row, 0, 0, 0, -1, -1, rr.orig};
            for (int r = 0; r < key.n_terms; r++)
                if (rr.cnt[r] > 0) { it.term = r; it.q_begin = rr.qb[r]; it.count = (int32_t)rr.cnt[r]; }
            if (key.small > 0 && it.count <= key.small) small.push_back(SmallItem{row, rr.orig, it.q_begin, it.count, 0});
            else if (lr_on && it.count <= key.lr) lr.push_back(SmallItem{row, rr.orig, it.q_begin, it.count, 0});
            else if (lr_on && it.count <= key.lr32) lr32.push_back(SmallItem{row, rr.orig, it.q_begin, it.count, 0});
            else direct.push_back(it);
        } else {
            SplitRow sr{row, (int32_t)split.size(), n_items, 0};
            for (int r = 0; r < key.n_terms; r++) {
                const int64_t beg = rr.qb[r], n = rr.cnt[r];
                const int pieces = (int)std::min<int64_t>((n + Tp - 1) / Tp, MAX_PIECES);
                for (int s = 0; s < pieces; s++) {
                    // equal pieces rather than T, T, ..., remainder
                    const int64_t b0 = beg + n * s / pieces, b1 = beg + n * (s + 1) / pieces;
                    split.push_back(Item{row, r, b0, (int32_t)(b1 - b0), (int32_t)split.size(), (int32_t)srows.size(), rr.orig});
                }
            }
            srows.push_back(sr);
        }
    }
    // launch order.  The items are listed longest first (split pieces, then rows by falling observation count); waves
    // that share a SIMD should be at different phases (the gather/MFMA phase of one under the factorisation of another),
    // so neighbours in launch order should differ in length: a fixed stride permutation of the sorted list.
    const int64_t total = (int64_t)split.size() + (int64_t)direct.size();
    std::vector<int32_t> order((size_t)total);
    {
        auto gcd = [](int64_t x, int64_t y) { while (y) { int64_t t = x % y; x = y; y = t; } return x; };
        int64_t stride = 1;
        if (total > 2) {
            stride = (int64_t)(0.6180339887 * (double)total) | 1;
            while (gcd(stride, total) != 1) stride += 2;
        }
        for (int64_t i = 0; i < total; i++) order[(size_t)i] = (int32_t)((i * stride) % total);
    }
    int rc;
    if (key.col > 0 && (rc = bdf_col_plan_build(ctx, crows, key.col, key.col_slots, plan.col))) return rc;
    plan.rows_lr = (int64_t)lr.size() + (int64_t)lr32.size(); plan.rows_small = (int64_t)small.size(); plan.rows_col = (int64_t)crows.size();
    plan.rows_k1 = (int64_t)direct.size() + (int64_t)srows.size();
    while (small.size() % 4) small.push_back(SmallItem{-1, 0, 0, 0, 0});
    plan.n_small = (int64_t)small.size();
    if (!small.empty() && (rc = to_device(small, &plan.small_dev))) return rc;
    plan.n_lr = (int64_t)lr.size() + (int64_t)lr32.size();
    if (plan.n_lr > 0) {
        // longest first: the waves of a workgroup then have rows of like length
        auto by_count = [](const SmallItem &x, const SmallItem &y) { return x.count > y.count; };
        std::stable_sort(lr.begin(), lr.end(), by_count);
        std::stable_sort(lr32.begin(), lr32.end(), by_count);
        std::vector<int32_t> lr_rows;
        lr_rows.reserve((size_t)plan.n_lr);
        for (const SmallItem &x : lr) lr_rows.push_back(x.row);
        for (const SmallItem &x : lr32) lr_rows.push_back(x.row);
        // (the positions in ASCENDING order: they are what the dense passes over the rows walk -- the back-transform x = L^-T q and
        // the per-row prior means -- and a pass over rows in the sampler's order, longest first, reads and writes 512-byte rows at
        // random)
        std::sort(lr_rows.begin(), lr_rows.end());
        while (lr.size() % 4) lr.push_back(SmallItem{-1, 0, 0, 0, 0});      // four rows per wave
        while (lr32.size() % 4) lr32.push_back(SmallItem{-1, 0, 0, 0, 0});
        plan.n_lr_padded = (int64_t)lr.size();
        plan.n_lr32_padded = (int64_t)lr32.size();
        lr.insert(lr.end(), lr32.begin(), lr32.end());
        if ((rc = to_device(lr, &plan.lr_dev)) || (rc = to_device(lr_rows, &plan.lr_rows_dev))) return rc;
    }
    if ((rc = to_device(direct, &plan.direct_dev)) || (rc = to_device(split, &plan.split_dev)) ||
        (rc = to_device(srows, &plan.rows_dev)) || (rc = to_device(order, &plan.order_dev)))
        return rc;
    BDF_HIP(hipMalloc((void **)&plan.partials_dev, std::max<size_t>(split.size() * (size_t)psz * sizeof(double), 8)));
    BDF_HIP(hipMalloc((void **)&plan.arrived_dev, std::max<size_t>(srows.size() * sizeof(int32_t), 8)));
    // on the launch stream: hipMemset runs on the NULL stream and returns before the device has done it, and a kernel on a
    // non-blocking stream does not wait for it -- the first launch of a new plan could have its counters zeroed under it
    // (a split row then never finds its last piece: the row keeps its old content)
    BDF_HIP(hipMemsetAsync(plan.arrived_dev, 0, std::max<size_t>(srows.size() * sizeof(int32_t), 8), ctx->stream));
    plan.dev.direct = plan.direct_dev; plan.dev.n_direct = (int32_t)direct.size();
    plan.dev.split = plan.split_dev;   plan.dev.n_split = (int32_t)split.size();
    plan.dev.rows = plan.rows_dev;     plan.dev.n_split_rows = (int32_t)srows.size();
    plan.dev.partials = plan.partials_dev;
    plan.dev.arrived = plan.arrived_dev;
    plan.dev.order = plan.order_dev;
    return BDF_OK;
}

// Lambda mu_i for MANY rows (per-row prior means: an entity with side information, macau.jl:104) -- one thread per output element with
// its row of Lambda in registers and the rows' means broadcast from LDS, where k_prior spends eight lanes and a butterfly on every
// element (100,000 rows at D = 32: 214 us -> ~20).  The same sums in the same order: the eight chains i = p, p + 8, ... by fma, then
// ((v0 + v4) + (v2 + v6)) + ((v1 + v5) + (v3 + v7)) -- the butterfly as k_prior's lane part 0 sees it.
template <int DPAD>
__global__ __launch_bounds__(256) void k_prior_rows(int D, int64_t nrows, const double *__restrict__ Lambda, const double *__restrict__ mu,
                                                    double *__restrict__ out_b)
{
    __shared__ double m_s[256];
    const int tid = threadIdx.x;
    const int RP = 256 / D;                                // rows per pass
    const int r = tid / D, e = tid - r * D;
    const bool mine = r < RP;
    double L[DPAD];
#pragma unroll
    for (int i = 0; i < DPAD; i++) L[i] = (mine && i < D) ? Lambda[e + (int64_t)i * D] : 0.0;
    for (int64_t row0 = (int64_t)blockIdx.x * RP; row0 < nrows; row0 += (int64_t)gridDim.x * RP) {
        const bool ok = mine && row0 + r < nrows;
        if (ok) m_s[tid] = mu[(row0 + r) * D + e];         // (tid == r * D + e: the pass's rows are contiguous)
        __syncthreads();
        if (ok) {
            const double *m = m_s + r * D;
            double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < DPAD; i++)
                if (i < D) v[i & 7] = fma(L[i], m[i], v[i & 7]);
            out_b[(row0 + r) * D + e] = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
        }
        __syncthreads();
    }
}

// which k_rows variant a launch takes: every term a two-mode relation on the lean gather path (matrix), and of those the
// launches with ONE relation whose values are coded (ratings)
void launch_kind(const SampleArgs &a, bool dump, bool &matrix, bool &coded, bool wide_ok = false)
{
    matrix = true;             // (wide_ok: D > 32, where the two-mode variant also takes factor matrices of 4 GiB or more -- lean == 2)
    for (int r = 0; r < a.n_terms; r++) matrix = matrix && (a.t[r].lean == 1 || (wide_ok && a.t[r].lean == 2)) && a.t[r].n_other == 1;
    static const bool no_matrix = getenv("BDF_K1_GENERAL_KERNEL") != nullptr;      // test hook: the general variant
    matrix = matrix && !no_matrix;
    static const bool no_coded = getenv("BDF_K1_NO_CODED") != nullptr;               // test hook: the uncoded two-mode variant
    coded = matrix && !dump && !no_coded && a.n_terms == 1 && a.t[0].lean == 1 && a.t[0].packed != nullptr && a.t[0].n_codes <= BDF_K1_CODES;
}

template <int DP>
int launch(bdf_ctx *ctx, const SampleArgs &a, Plan &plan, bool dump)
{
    constexpr int WPB = Geo<DP>::WPB;
    PlanDev p = plan.dev;
    bool matrix, coded;
    launch_kind(a, dump, matrix, coded, DP == 64);
    const int64_t waves = (int64_t)p.n_split + p.n_direct;
    if (waves > 0) {
        const dim3 grid((unsigned)((waves + WPB - 1) / WPB)), block(64 * WPB);
        auto kern = dump ? (matrix ? k_rows<DP, true, true> : k_rows<DP, true, false>)
                         : (coded ? k_rows<DP, false, true, true> : (matrix ? k_rows<DP, false, true> : k_rows<DP, false, false>));
        // start / stop events (bdf_ctx_time_next_rows) ride on the dispatch packet itself: the kernel's own begin and end,
        // no marker packets around it
        hipExtLaunchKernelGGL(kern, grid, block, 0, ctx->stream, dump ? nullptr : ctx->time_start, dump ? nullptr : ctx->time_stop, 0, a, p);
        if (!dump) ctx->time_start = ctx->time_stop = nullptr;
        BDF_HIP(hipGetLastError());
    }
    return BDF_OK;
}

}  // namespace

// parity hook: split rows whose pieces did not all arrive in the launches so far (their arrival counters reset themselves
// when the last piece arrives, so any non-zero counter after a completed launch is a row that was never finished)
extern "C" int bdf_rows_unfinished(bdf_ctx *ctx, int64_t *count)
{
    BDF_REQUIRE(ctx && count, BDF_ERR_ARG, "bdf_rows_unfinished: NULL argument");
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    *count = 0;
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto it = g_caches.find(ctx);
    if (it == g_caches.end()) return BDF_OK;
    for (auto &kv : it->second.plans) {
        if (kv.second.col.n_split_rows > 0) {
            std::vector<int32_t> hc((size_t)kv.second.col.n_split_rows);
            BDF_HIP(hipMemcpy(hc.data(), kv.second.col.arrived_dev, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (int32_t v : hc) *count += v != 0;
        }
        const int n = kv.second.dev.n_split_rows;
        if (n <= 0) continue;
        std::vector<int32_t> h((size_t)n);
        BDF_HIP(hipMemcpy(h.data(), kv.second.arrived_dev, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int32_t v : h) *count += v != 0;
        static const bool dbg = getenv("BDF_DEBUG_UNFINISHED") != nullptr;
        if (dbg)
            for (int i = 0; i < n; i++)
                if (h[(size_t)i] != 0)
                    fprintf(stderr, "[bdf] unfinished: plan DP=%d T=%d Tp=%d shard %d/%d terms %d, split row %d of %d: counter %d (array %p)\n",
                            kv.first.DP, kv.first.T, kv.first.Tp, kv.first.shard, kv.first.n_shards, kv.first.n_terms, i, n, h[(size_t)i],
                            (void *)kv.second.arrived_dev);
    }
    return BDF_OK;
}

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
            if (kv->second.small_dev) (void)hipFree(kv->second.small_dev);
            if (kv->second.lr_dev) (void)hipFree(kv->second.lr_dev);
            if (kv->second.lr_rows_dev) (void)hipFree(kv->second.lr_rows_dev);
            (void)hipFree(kv->second.partials_dev); (void)hipFree(kv->second.arrived_dev); (void)hipFree(kv->second.order_dev);
            bdf_col_plan_free(kv->second.col);
            kv = plans.erase(kv);
        } else {
            ++kv;
        }
    }
    if (rel_serial == 0) g_caches.erase(it);
}

// Lambda mu (D doubles) and the accumulator-layout image of the index-reversed Lambda (k_prior), for k_block.hip
int bdf_prior_image(bdf_ctx *ctx, int D, const double *Lambda, const double *mu, double *out_b, double *out_c)
{
    const int DPp = D <= 16 ? 16 : (D <= 32 ? 32 : 64);
    const int nimg = (DPp / 16) * (DPp / 16 + 1) / 2 * 4;
    const int64_t waves = ((int64_t)D + 7) / 8 + nimg;
    hipLaunchKernelGGL(k_prior, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, ctx->stream, D, DPp, (int64_t)1, Lambda, mu, 0, out_b, out_c);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

int bdf_launch_sample_rows(bdf_ctx *ctx, const SampleArgs &a_in, const bdf_rel *const *rels, const int *modes, int shard,
                           int n_shards, bool dump)
{
    SampleArgs a = a_in;
    if (a.prior_b == nullptr) {         // no prior pack from bdf_hyper_sample: derive Lambda mu and the image here
        // prior part of b: Lambda mu (one vector) or Lambda mu_i for every row (per-row prior means, macau.jl:104)
        const int64_t N = rels[0]->nint[modes[0]];
        const int64_t nr = a.mu_is_matrix ? N : 1;
        const int DPp = a.D <= 16 ? 16 : (a.D <= 32 ? 32 : 64);
        const int nimg = (DPp / 16) * (DPp / 16 + 1) / 2 * 4;          // (block, register) pairs of the image
        void *pb;
        int rc = bdf_scratch(ctx, ((size_t)nr * a.D + (size_t)nimg * 64) * sizeof(double), &pb);
        if (rc) return rc;
        static const bool no_prior_rows = getenv("BDF_PRIOR_ROWS") && atoi(getenv("BDF_PRIOR_ROWS")) == 0;     // test hook: k_prior for every size
        if (a.mu_is_matrix && nr >= 4096 && !no_prior_rows) {
            // many rows: Lambda mu_i by k_prior_rows (the same sums in the same order), the image alone by k_prior (nrows = 0)
            const int RP = 256 / a.D;
            const unsigned grid = (unsigned)std::min<int64_t>((nr + RP - 1) / RP, 4096);
            if (DPp == 16) hipLaunchKernelGGL(k_prior_rows<16>, dim3(grid), dim3(256), 0, ctx->stream, a.D, nr, a.Lambda, a.mu, (double *)pb);
            else if (DPp == 32) hipLaunchKernelGGL(k_prior_rows<32>, dim3(grid), dim3(256), 0, ctx->stream, a.D, nr, a.Lambda, a.mu, (double *)pb);
            else hipLaunchKernelGGL(k_prior_rows<64>, dim3(grid), dim3(256), 0, ctx->stream, a.D, nr, a.Lambda, a.mu, (double *)pb);
            hipLaunchKernelGGL(k_prior, dim3((unsigned)((nimg + 3) / 4)), dim3(256), 0, ctx->stream, a.D, DPp, (int64_t)0, a.Lambda, a.mu,
                               0, (double *)pb, (double *)pb + nr * a.D);
        } else {
            const int64_t waves = (nr * a.D + 7) / 8 + nimg;
            hipLaunchKernelGGL(k_prior, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, ctx->stream, a.D, DPp, nr, a.Lambda, a.mu,
                               a.mu_is_matrix, (double *)pb, (double *)pb + nr * a.D);
        }
        BDF_HIP(hipGetLastError());
        a.prior_b = (const double *)pb;
        a.prior_c = (const double *)pb + nr * a.D;
    }
    const int DP = a.D <= 16 ? 16 : (a.D <= 32 ? 32 : 64);
    const int DB = DP / 16, NB = DB * (DB + 1) / 2;
    const int psz = NB * 4 * 64 + DB * 16;
    PlanKey key;
    memset(&key, 0, sizeof(key));
    for (int r = 0; r < a.n_terms; r++) { key.rel[r] = rels[r]->serial; key.mode[r] = modes[r]; }
    key.n_terms = a.n_terms; key.DP = DP; key.T = ctx->item_size; key.Tp = std::min(ctx->piece_size, ctx->item_size); key.shard = shard; key.n_shards = n_shards;
    if (ctx->item_auto) {
        // Rows are cut into pieces so that a launch of a few thousand rows has no wave much longer than the others.  A launch with
        // hundreds of waves per resident slot has no such tail, and every piece costs a partial sum written to the slab and read
        // back (21 KB at D = 64: the 540,000 pieces of configuration C4's item launch moved 22 GB): larger items there -- about
        // sixteen waves per slot, between the default and 2048 observations (the same for every shard of the launch).
        // (a NOMINAL slot count -- 256 CUs -- not the device's or the stream's: the cut of a row, and with it the order of its
        // floating-point sums, must not depend on the CU count or on BDF_RESERVE_CUS)
        int64_t nnz_launch = 0;
        for (int r = 0; r < a.n_terms; r++) nnz_launch += rels[r]->idx[modes[r]].own_nnz;
        const int64_t slots = (int64_t)256 * 4 * (DP == 64 ? BDF_K1_WAVES64 : (DP == 32 ? BDF_K1_WAVES32C : 8));
        const int64_t t = std::min<int64_t>(2048, (nnz_launch / (slots * 16) + 63) / 64 * 64);
        if (t > key.T) { key.T = (int)t; key.Tp = (int)(t * 2 / 3); }
    }
    // D <= 16, one two-mode relation with the lean gather and no per-observation baseline, an entity of many rows: its short
    // rows four to a wave (k_rows_small).  bdf_ctx_set_small_rows: the longest row taken that way (default 48 observations,
    // environment BDF_K1_SMALL; 0: off) and the smallest entity (default 8192 rows, BDF_K1_SMALL_MIN_ROWS: below that the
    // second launch costs more than it saves)
    {
        const int small_max = ctx->small_max;
        const int64_t small_rows = ctx->small_min_rows;
        const int64_t n_rows_all = rels[0]->sharded ? (int64_t)rels[0]->idx[modes[0]].own_orig.size() : (int64_t)rels[0]->idx[modes[0]].order.size();
        if (DP == 16 && !dump && small_max > 0 && a.n_terms == 1 && a.t[0].lean == 1 && a.t[0].n_other == 1 && a.t[0].linear == nullptr &&
            n_rows_all >= small_rows)
            key.small = std::min(small_max, ctx->item_size);
    }

    // D > 16, one two-mode relation without per-observation baselines (shared or per-row prior means): the rows of few observations
    // by the low-rank sampler (k_rows_lr.hip; bdf_ctx_set_lowrank, environment BDF_LOWRANK:
    // the longest such row, -1 = min(16, D / 2), 0 = off) -- when there are enough of them (decided when the plan is built)
    int64_t M_other = 0;
    if (DP > 16 && !dump && ctx->lr_max != 0 && a.n_terms == 1 && a.t[0].n_other == 1 && a.t[0].linear == nullptr) {
        const int other = 1 - modes[0];
        M_other = rels[0]->nint[other];
        const int lr_want = ctx->lr_max < 0 ? a.D / 2 : ctx->lr_max;
        key.lr = std::min(std::min(lr_want, bdf_lr_max_observations()), ctx->item_size);
        key.lr32 = DP == 64 ? std::min(std::min(lr_want, bdf_lr32_max_observations()), ctx->item_size) : 0;       // (> key.lr: rows of 17 .. 32 observations too)
        key.lr_min = std::max<int64_t>(ctx->lr_min_rows, 1);
        key.lr_other = ctx->lr_min_rows > 0 ? rels[0]->dims[other] : 0;          // (min_rows = 0, a test hook: whenever the entity has such a row)
    }

    // 16 < D <= 32, one two-mode relation on the lean gather path without per-observation baselines: the rows four to a wave in
    // the column layout (K1c, k_rows_col.hip; bdf_ctx_set_col_rows) -- unless the caller chose K1's item size or its general variant
    static const bool no_col = getenv("BDF_K1_GENERAL_KERNEL") != nullptr;          // (test hook: k_rows' general variant)
    if (DP == 32 && a.D > 16 && !dump && ctx->col_piece > 0 && (ctx->col_explicit || ctx->item_auto) && a.n_terms == 1 && a.t[0].n_other == 1 &&
        a.t[0].lean == 1 && a.t[0].linear == nullptr && !no_col) {
        static int cus = 0;
        if (!cus) {
            hipDeviceProp_t prop;
            BDF_HIP(hipGetDeviceProperties(&prop, ctx->device));
            cus = prop.multiProcessorCount;
        }
        static const int per_simd = getenv("BDF_COL_PER_SIMD") ? std::max(1, atoi(getenv("BDF_COL_PER_SIMD"))) : 2;
        key.col = ctx->col_piece;
        if (!ctx->col_explicit) {
            // A row of more than 4 T observations SPANS waves: every part writes its 6.4 KB of sums through to the slab and the part
            // that arrives last adds them, slot after slot -- ~25 us of a wave's slot per part when thousands of them are in flight
            // (profiles/r05_k1c_piece_size.txt: 1,000 rows of 15,000 observations, the reference's benchmark shape, 2.7 ms at
            // T = 128 in 30,000 parts, 0.70 ms at T = 1,024 in 4,000; 4,000 rows of 3,000: 0.73 -> 0.44 ms).  Small pieces are
            // for launches of ONE generation of waves (MovieLens: the heaviest wave is the launch's tail); a launch with many
            // waves per slot takes larger ones: about eight waves' worth of observations per slot of a NOMINAL 2,048 (not the
            // device's or the stream's: the cut of a row must not depend on them), from the WHOLE entity's count -- the same on
            // every shard, chunk and rank -- between the default and 2,048.
            const int64_t nnz_entity = (int64_t)rels[0]->idx[modes[0]].rowptr.back();
            const int64_t t = std::min<int64_t>(2048, (nnz_entity / (2048 * 8) + 63) / 64 * 64);
            if (t > key.col) key.col = (int)t;
        }
        key.col_slots = std::max(1, cus - ctx->reserve_cus) * 4 * per_simd;
        M_other = rels[0]->nint[1 - modes[0]];
    }

    Plan *plan;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        PlanCache &cache = g_caches[ctx];
        auto it = cache.plans.find(key);
        if (it == cache.plans.end()) {
            std::vector<RowRef> rows;
            if (rels[0]->sharded) {
                // a relation created with a layout holds this rank's rows only, chunk after chunk: `shard` is the chunk
                const bdf_mode_index &ix0 = rels[0]->idx[modes[0]];
                for (int64_t o = ix0.chunk_begin[(size_t)shard]; o < ix0.chunk_begin[(size_t)shard + 1]; o++) {
                    RowRef rr;
                    rr.out = ix0.own_pos[(size_t)o]; rr.orig = ix0.own_orig[(size_t)o];
                    for (int r = 0; r < a.n_terms; r++) {
                        const bdf_mode_index &ix = rels[r]->idx[modes[r]];
                        rr.qb[r] = ix.own_q[(size_t)o]; rr.cnt[r] = ix.own_q[(size_t)o + 1] - ix.own_q[(size_t)o];
                    }
                    rows.push_back(rr);
                }
            } else {
                // rows of this shard: positions shard, shard + n_shards, ... of the degree-descending order of the first
                // relation (the reference deals rows i:P:N to its P workers for the same balance, sampling.jl:154)
                const std::vector<int32_t> &order = rels[0]->idx[modes[0]].order;
                for (size_t pos = (size_t)shard; pos < order.size(); pos += (size_t)n_shards) {
                    RowRef rr;
                    rr.out = rr.orig = order[pos];
                    for (int r = 0; r < a.n_terms; r++) {
                        const auto &rp = rels[r]->idx[modes[r]].rowptr;
                        rr.qb[r] = rp[(size_t)rr.orig]; rr.cnt[r] = rp[(size_t)rr.orig + 1] - rp[(size_t)rr.orig];
                    }
                    rows.push_back(rr);
                }
            }
            // the low-rank sampler pays its set-up (the opposite factor transformed, two more launches) only with enough rows:
            // counted over the WHOLE entity (the host's index is the whole relation's on every rank), so that shards, chunks and
            // ranks decide alike
            bool lr_on = false;
            if (key.lr > 0) {
                const std::vector<int64_t> &rp = rels[0]->idx[modes[0]].rowptr;
                int64_t cnt = 0;
                for (size_t i = 0; i + 1 < rp.size(); i++) cnt += rp[i + 1] - rp[i] <= key.lr;
                lr_on = cnt >= key.lr_min && 2 * cnt >= key.lr_other;
            }
            Plan np;
            int rc = build_plan(ctx, key, rows, psz, lr_on, np);
            if (rc) return rc;
            it = cache.plans.emplace(key, np).first;
        }
        plan = &it->second;
    }
    {
        // bdf_ctx_rows_dispatch: the chunks / shards of one iteration's launch of the entity add up
        if (!ctx->rows_dispatch) ctx->rows_dispatch = new std::map<uint32_t, std::array<int64_t, 7>>();
        std::array<int64_t, 7> &rdsp = (*ctx->rows_dispatch)[a.entity_tag];
        if (rdsp[0] != (int64_t)a.sweep + 1) rdsp = {(int64_t)a.sweep + 1, 0, 0, 0, 0, 0, 0};
        rdsp[1] += plan->rows_lr; rdsp[2] += plan->rows_small; rdsp[3] += plan->rows_col; rdsp[4] += plan->rows_k1;
        rdsp[5] += (int64_t)plan->dev.n_split + plan->dev.n_direct; rdsp[6] += plan->col.n_waves;
    }
    if (plan->n_lr > 0) {
        const bool more = (int64_t)plan->dev.n_split + plan->dev.n_direct + plan->col.n_waves > 0;
        // the constants of the launch (L, the opposite factor transformed): once per entity launch -- a later chunk of the same
        // launch (same inputs, same iteration) finds them in the context
        const bool same = shard > 0 && ctx->lr_key_fac == (const void *)a.t[0].fac[0] && ctx->lr_key_Lambda == (const void *)a.Lambda &&
                          ctx->lr_key_mu == (const void *)a.mu && ctx->lr_key_sweep == a.sweep && ctx->lr_key_tag == a.entity_tag &&
                          ctx->lr_key_D == a.D && ctx->lr_key_M == M_other;
        int rc = bdf_lr_launch(ctx, a, M_other, rels[0]->nint[modes[0]], plan->lr_dev, plan->n_lr, plan->n_lr_padded, plan->n_lr32_padded, plan->lr_rows_dev, !same,
                               ctx->time_start, more ? nullptr : ctx->time_stop);
        if (rc) return rc;
        ctx->lr_key_fac = a.t[0].fac[0]; ctx->lr_key_Lambda = a.Lambda; ctx->lr_key_mu = a.mu; ctx->lr_key_sweep = a.sweep;
        ctx->lr_key_tag = a.entity_tag; ctx->lr_key_D = a.D; ctx->lr_key_M = M_other;
        ctx->time_start = nullptr;
        if (!more) { ctx->time_stop = nullptr; return BDF_OK; }
    }
    if (plan->n_small > 0) {
        // the short rows first (most of the entity), then k_rows for the others; a caller's timing events and the hand-over of
        // the draw stay with k_rows when it has anything to do
        const bool more = (int64_t)plan->dev.n_split + plan->dev.n_direct + plan->col.n_waves > 0;
        const dim3 grid((unsigned)((plan->n_small + 15) / 16)), block(256);
        hipEvent_t e0 = ctx->time_start, e1 = more ? nullptr : ctx->time_stop;
        const bool coded = a.t[0].packed != nullptr;
        const SmallItem *si = plan->small_dev;
        const int64_t ns = plan->n_small;
#define SMALL_LAUNCH(DRV)                                                                                                     \
        do {                                                                                                                    \
            if (coded) hipExtLaunchKernelGGL((k_rows_small<true, DRV>), grid, block, 0, ctx->stream, e0, e1, 0, a, si, ns);     \
            else hipExtLaunchKernelGGL((k_rows_small<false, DRV>), grid, block, 0, ctx->stream, e0, e1, 0, a, si, ns);          \
        } while (0)
        if (a.D <= 4) SMALL_LAUNCH(4); else if (a.D <= 8) SMALL_LAUNCH(8); else if (a.D <= 12) SMALL_LAUNCH(12); else SMALL_LAUNCH(16);
#undef SMALL_LAUNCH
        BDF_HIP(hipGetLastError());
        ctx->time_start = nullptr;
        if (!more) { ctx->time_stop = nullptr; return BDF_OK; }
    }
    if (plan->col.n_waves > 0) {
        const bool more = (int64_t)plan->dev.n_split + plan->dev.n_direct > 0;
        static const bool no_coded = getenv("BDF_K1_NO_CODED") != nullptr;               // test hook: ids and values instead of the packed words
        SampleArgs ac = a;
        if (no_coded) ac.t[0].packed = nullptr;
        // the rows' hand-over by counter (SampleArgs::done): only when this launch is ALL of the call's rows
        if (more || plan->n_lr > 0 || plan->n_small > 0) ac.done = nullptr;
        if (ac.done) ctx->rows_done_added = plan->col.n_waves;
        int rc = bdf_col_launch(ctx, ac, plan->col, M_other, ctx->time_start, more ? nullptr : ctx->time_stop);
        if (rc) return rc;
        ctx->time_start = nullptr;
        if (!more) { ctx->time_stop = nullptr; return BDF_OK; }
    }
    if (DP == 16) return launch<16>(ctx, a, *plan, dump);
    if (DP == 32) return launch<32>(ctx, a, *plan, dump);
    return launch<64>(ctx, a, *plan, dump);
}
