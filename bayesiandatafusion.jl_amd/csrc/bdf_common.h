// bdf_common.h -- internal declarations shared by the HIP translation units of libbdf_hip.so
#pragma once
#include <map>
#include <array>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <vector>
#include "../../include/bdf.h"

void bdf_set_error(const char *fmt, ...);

#define BDF_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            bdf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return BDF_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define BDF_REQUIRE(cond, code, ...)                                                       \
    do {                                                                                   \
        if (!(cond)) { bdf_set_error(__VA_ARGS__); return (code); }                        \
    } while (0)

struct bdf_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    uint64_t seed;
    uint32_t *sweep_dev;       // Gibbs iteration counter in device memory
    uint32_t sweep_host;       // mirror of the last bdf_ctx_set_sweep / advance
    // scratch (grown on demand, never shrunk)
    void *scratch;
    size_t scratch_bytes;
    void *scratch2;            // second block: split-K partials of the dense products (used while `scratch` is held)
    size_t scratch2_bytes;
    int small_max;             // k_rows_small: longest row (observations) sampled four to a wave at D <= 16; 0: off
    int64_t small_min_rows;    // ... and the smallest entity (rows) for which it is used
    // k_rows_lr (k_rows_lr.hip): rows of few observations sampled by the low-rank map instead of the reference's
    int lr_max;                // longest row (observations) sampled that way at D > 16; -1: min(16, D / 2); 0: off
    int64_t lr_min_rows;       // ... and the smallest number of such rows in a launch for which it is used
    int col_piece;             // K1c (k_rows_col.hip): one two-mode relation at 16 < D <= 32 four rows per wave in the column layout, rows cut
                               // into pieces of at most this many observations; 0: off
    bool col_explicit;         // ... set by the caller (bdf_ctx_set_col_rows): taken also when the caller chose K1's item size
    double *lr_T;              // Tf | Tb | Tm (64 x 64 each) | L' mu (64): the launch's constants (k_lr_prep)
    double *lr_mrows;          // per-row prior means transformed (L' mu_i, rows of DP doubles), grown on demand
    size_t lr_mrows_bytes;
    double *lr_vt;             // the opposite entity's factor matrix transformed (V L^-T), grown on demand
    size_t lr_vt_bytes;
    // what lr_T / lr_vt were computed from: a later chunk of the same entity launch reuses them
    const void *lr_key_fac, *lr_key_Lambda, *lr_key_mu; uint32_t lr_key_sweep, lr_key_tag; int lr_key_D; int64_t lr_key_M;
    int *flag_dev;             // not-positive-definite flag (bits 1..32: errors; 64: BDF_WARN_CG_MAXITER): the device address of
    int *flag_host;            // ... a word of mapped, coherent HOST memory (kernels atomicOr into it on their error paths only;
                               // bdf_ctx_sync reads it without a copy -- a 4-byte blocking device-to-host copy is ~10 us)
    uint32_t warnings;         // non-fatal bits seen by bdf_ctx_sync, until bdf_ctx_warnings takes them
    int item_size;             // K1: observations per work item (rows longer than this are split)
    int piece_size;            // K1: ... into pieces of at most this many observations
    bool item_auto;            // K1: neither was set by the caller: large launches take larger items (bdf_launch_sample_rows)
    int gather_mode;           // K1 parity hook: 0 auto, 1 general gather path, 2 lean path with 64-bit row offsets (D > 32)
    hipEvent_t time_start, time_stop;      // bdf_ctx_time_next_rows: attached to the next row-kernel dispatch, then cleared
    unsigned long long *rows_span;         // bdf_ctx_span_next_rows: SampleArgs::span of the next row launch, then cleared
    const uint32_t *rows_ready;            // (library-internal) SampleArgs::ready of the next row launch, then cleared
    uint32_t rows_ready_want;
    // (library-internal, bdf_gibbs_sweep) the rows' hand-over to the hyperprior chain WITHOUT an event: when the next bdf_sample_rows is
    // one k_rows_col launch, every wave of it -- its rows written through, the stores drained -- adds 1 to shard (wave % 64) of these
    // 64 words (16 words apart), and rows_done_added says how many will; else rows_done_added stays -1 and the caller uses the event
    uint32_t *rows_done;
    int64_t rows_done_added;
    const uint32_t *hyper_wait;            // (library-internal) ... and the next one-launch chain (k_hyper_chain) polls their sum for this target
    uint32_t hyper_wait_target;
    uint32_t *hyper_ready;                 // (library-internal) word the next bdf_hyper_sample sets to hyper_ready_value once its pack is written, then cleared
    uint32_t hyper_ready_value;
    hipEvent_t time_h_start, time_h_stop;  // bdf_ctx_time_next_hyper: start of the next sums kernel, end of the next draw kernel
    // batched CG (k_feat.hip): device flag that lets product kernels enqueued ahead return at once (NULL outside a solve),
    // and the host-mapped words through which the device reports (iteration, active columns)
    // CU reservation (bdf_ctx_create_rows): the last... the first `reserve_cus` bits of the CU mask (one CU per XCD each 8)
    // are kept free of this context's kernels, for the side context created with reserved = 1
    int reserve_cus;           // CUs set aside by the row context this one belongs to (0: none)
    int on_reserved;           // this context's stream runs on the reserved CUs only
    const int *skip_flag;
    volatile uint64_t *cg_status;
    // bdf_gibbs_sweep: the next bdf_hyper_sums leaves its second stage to the bdf_hyper_sample that follows it on this context
    bool hyper_fuse;
    const double *hyper_partial; int hyper_nblocks; double *hyper_sumU, *hyper_UUt;
    bool hyper_chain;                   // ... and launches nothing itself: the draw's launch carries the sums' workgroups (k_hyper_chain)
    int hyper_chain_D; int64_t hyper_chain_N, hyper_chain_rpb; const double *hyper_chain_sample, *hyper_chain_uhat;
    unsigned *hyper_count;              // partial workgroups finished (k_hyper_chain), allocated at first use
    double *hyper_chain_draws;          // (bdf_gibbs_sweep) the chain's launch also makes the entity's random part (k_hyper_draws' values) into this buffer
    double *cg_part;                    // partial dot products of the chunked CG step (k_cg_long_*), allocated at first use
    uint32_t cg_gen;
    unsigned *cg_bar;                   // the hand-over counter of the one-launch CG solve (k_cg_resident), allocated at first use
    // bdf_ctx_rows_dispatch: per entity tag {iteration number, rows by K1-lr, K1s, K1c, K1, K1's items, K1c's waves} of the latest launch
    std::map<uint32_t, std::array<int64_t, 7>> *rows_dispatch;
};

int bdf_scratch(bdf_ctx *ctx, size_t bytes, void **out);
int bdf_scratch2(bdf_ctx *ctx, size_t bytes, void **out);
int bdf_sum_ranks_into(bdf_ctx *ctx, bdf_comm *comm, double *x, int64_t n, double *gather);
int bdf_sample_beta_rel_impl(bdf_ctx *ctx, bdf_comm *comm, const bdf_feat *fc, const bdf_pairs *train, int64_t first_obs, int D,
                             const double *const *factors, double mean_value, double alpha, const double *alpha_dev, double lambda_beta,
                             uint32_t rel_tag, double *beta_out, double *linear_out, double *rhs_out);
#define BDF_DRAWS_BATCH 8
int bdf_hyper_draws_batch(bdf_ctx *ctx, int D, int n, const int64_t *N, const double *nu, const uint32_t *entity_tag, double *const *draws_out);

struct bdf_mode_index {
    std::vector<int64_t> rowptr;   // host, dims+1
    std::vector<int64_t> rowids;   // host, nnz, 1-based COO row numbers (IndexedDF.index)
    std::vector<int32_t> order;    // host, rows by descending degree (stable)
    int64_t *rowptr_dev;           // dims+1
    int32_t *colidx_dev;           // (n_modes-1) planes of nnz: 0-based ids of the other modes, mode order
    double  *vals_dev;             // nnz values in mode order
    int32_t *perm_dev;             // nnz: 0-based COO row number in mode order
    uint32_t *packed_dev;          // nullable, two-mode relations with <= 256 distinct values: (value code << 24) | other-mode id
    int32_t *order_dev;
    // relation created with a layout (bdf_relation_create_sharded): the device arrays hold only the observations of the rows
    // this rank owns, chunk after chunk in internal-position order, and colidx holds INTERNAL positions of the other modes
    std::vector<int32_t> own_orig;     // original id of every owned row
    std::vector<int32_t> own_pos;      // its internal position (row of the factor matrix)
    std::vector<int64_t> own_q;        // offsets of the owned rows' observations in the device arrays (owned + 1)
    std::vector<int64_t> chunk_begin;  // first owned row of every chunk (chunks + 1)
    int64_t own_nnz;
};

struct bdf_rel {
    bdf_ctx *ctx;
    uint64_t serial;           // unique per created relation (keys the K1 plan cache)
    int n_modes;
    int64_t dims[BDF_MAX_MODES];
    int64_t nnz;
    double value_mean;
    bdf_mode_index idx[BDF_MAX_MODES];
    int sharded;                       // created with a layout
    int rank, world, chunks;
    int64_t nint[BDF_MAX_MODES];       // rows of the factor matrix of every mode (== dims without a layout)
    // ratings take few distinct values (MovieLens: 5): when there are at most 256 the relation also keeps them as 8-bit codes
    // into this table, packed with the other mode's id (K1's coded two-mode variant: no value registers in its pipeline)
    int n_codes;                       // 0: not coded
    double *table_dev;                 // 256 doubles, ascending distinct values (the rest zero)
};

struct bdf_pairs {
    bdf_ctx *ctx;
    int n_modes;
    int64_t n;
    int32_t *ids_dev;     // n_modes planes of n, 0-based
    double *values_dev;   // n
    double *avg_dev, *sq_dev;
    double count;         // counter_prob (macau.jl:171-183)
    const double *baseline_dev;   // nullable, borrowed: per-pair baseline replacing mean_value (relation features)
    int32_t *orig_dev;            // nullable: storage position -> caller's index (bdf_pairs_sort)
    int sorted_mode;              // the mode bdf_pairs_sort sorted by, -1: caller's order
    std::vector<int32_t> ids_host, orig_host;
    std::vector<double> values_host;
};

struct bdf_feat {
    bdf_ctx *ctx;
    int kind;             // 0 dense, 1 csr (real), 2 binary
    int64_t m, n, nnz;
    double *dense_dev;    // m x n column-major
    // CSR (rows) and CSC (= CSR of F') so that neither product needs atomics
    int64_t *rowptr_dev; int32_t *colind_dev; double *rvals_dev;
    int64_t *colptr_dev; int32_t *rowind_dev; double *cvals_dev;
    // column panels of the sparse products (k_feat.hip, spmm): entries of row r with a column in panel p are
    // [panel_ptr[p * rows + r], panel_ptr[(p + 1) * rows + r]) -- NULL when the operand is small or a row's entries are not in column order
    int64_t *panel_fwd_dev; int n_panels_fwd;      // F   (rows m, panels over the n columns)
    int64_t *panel_tr_dev; int n_panels_tr;        // F'  (rows n, panels over the m columns)
    double *FF_dev;       // n x n (F'F), built on first use_ff
    int32_t *row_ids_dev; // nullable (bdf_feat_set_row_ids): original id of every row of F, keys the rows' noise streams
    double *chol_ws;      // workspace of the blocked direct solve (k_chol.hip), allocated on first use
    size_t chol_ws_doubles;
    void *gather_dev;     // several ranks: the blocks of beta columns the ranks exchange (bdf_sample_beta_ranks), on first use
    size_t gather_bytes;
    // direct solve of a mid-sized feature set (64 < n <= BDF_EIG_MAX): F'F = Q diag(s) Q' once (host, first use), then
    // (F'F + lambda I) \ rhs = Q ((Q' rhs) ./ (s + lambda)) per iteration -- lambda changes every iteration, Q and s do not
    bdf_feat *eig_Q;      // Q as a dense n x n operator (eigenvectors in columns)
    double *eig_s;        // n eigenvalues (dev)
    double *eig_y;        // n x D workspace (dev), eig_y_cols columns
    int eig_y_cols;
    bool eig_failed;      // the decomposition's QL iteration did not converge: this operator takes the factorisation (bdf_chol_solve) instead
    bool FF_summed;       // several ranks, F split by rows (bdf_sample_beta_rel_ranks): FF_dev already holds the sum over the ranks
};
#define BDF_EIG_MAX 640

// (FF + lambda I) \ rhs for all D right-hand sides by blocked Cholesky (solve_full, src/sampling.jl:314-320)
int bdf_chol_solve(bdf_ctx *ctx, bdf_feat *f, int D, const double *lambda_dev, const double *rhs, double *beta);

// ---------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller (host+device).  Counter layout: DESIGN.md "RNG contract".
// ---------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__host__ __device__ __forceinline__ u32x4 bdf_draw(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
                                          uint64_t row, uint32_t pair)
{
    u32x4 c;
    c.x = (uint32_t)row;
    c.y = (uint32_t)((row >> 32) & 0xffffu) | (pair << 16);
    c.z = sweep;
    c.w = (purpose << 24) | (entity & 0xffffffu);
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

__host__ __device__ __forceinline__ double bdf_u01(uint32_t lo, uint32_t hi)
{
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * 0x1.0p-53;
}

// sin(2 pi u), cos(2 pi u) for u in (0, 1).  The argument is reduced in u, exactly: q = nearest integer to 4u, r = u - q/4
// in [-1/8, 1/8]; then the fdlibm kernel polynomials on |2 pi r| <= pi/4 and the quadrant.  Within ~2e-16 of sin / cos of the
// rounded product 2 pi u (the oracle's libm calls) at a seventh of the instructions: the library routines carry a
// Payne-Hanek path and double-double arithmetic for arguments this code never sees, and Box-Muller was a quarter of the
// row kernel's VALU instructions.
__device__ __forceinline__ void bdf_sincos2pi(double u, double &s, double &c)
{
    const double q = rint(4.0 * u);
    const double x = 6.283185307179586476925286766559 * fma(q, -0.25, u);
    const double z = x * x;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                   2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                     8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                   -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                     -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double sx = fma(x * z, ps, x);
    const double cx = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int iq = (int)q & 3;
    const double sa = (iq & 1) ? cx : sx, ca = (iq & 1) ? sx : cx;
    s = (iq & 2) ? -sa : sa;
    c = (iq == 1 || iq == 2) ? -ca : ca;
}

// log(x) for a normal x in (0, 1) -- bdf_u01 never returns 0, 1 or a denormal: the fdlibm algorithm (x = 2^k m, m in
// [sqrt(2)/2, sqrt(2)), log m from s = f / (2 + f), f = m - 1), < 1 ulp, without the library routine's special cases
__device__ __forceinline__ double bdf_log01(double x)
{
    int k;
    double m = frexp(x, &k);                        // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }
    const double f = m - 1.0, dk = (double)k;
    const double s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

// standard normal number `n` (0-based) of stream (purpose, entity, row): pair n/2, element n%2
__device__ __forceinline__ double bdf_normal(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
                                    uint64_t row, int n)
{
    u32x4 o = bdf_draw(seed, sweep, purpose, entity, row, (uint32_t)(n >> 1));
    double u1 = bdf_u01(o.x, o.y), u2 = bdf_u01(o.z, o.w);
    double r = sqrt(-2.0 * bdf_log01(u1));
    double s, c;
    bdf_sincos2pi(u2, s, c);
    return (n & 1) ? r * s : r * c;
}

// both normals of pair `pair` of the stream: numbers 2 pair and 2 pair + 1 (the same values bdf_normal returns)
__device__ __forceinline__ void bdf_normal_pair(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
                                       uint64_t row, uint32_t pair, double &z0, double &z1)
{
    u32x4 o = bdf_draw(seed, sweep, purpose, entity, row, pair);
    double u1 = bdf_u01(o.x, o.y), u2 = bdf_u01(o.z, o.w);
    double r = sqrt(-2.0 * bdf_log01(u1));
    double s, c;
    bdf_sincos2pi(u2, s, c);
    z0 = r * c;
    z1 = r * s;
}

__device__ __forceinline__ double bdf_uniform(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
                                     uint64_t row, uint32_t pair)
{
    u32x4 o = bdf_draw(seed, sweep, purpose, entity, row, pair);
    return bdf_u01(o.x, o.y);
}

// Gamma(a, 1), Marsaglia-Tsang; variate index g addresses the stream, the attempt is the pair
__device__ __forceinline__ double bdf_gamma(uint64_t seed, uint32_t sweep, uint32_t entity, uint64_t g, double a)
{
    double boost = 1.0;
    if (a < 1.0) {
        boost = pow(bdf_uniform(seed, sweep, BDF_P_GAMMA_U, entity, g, 0xffffu), 1.0 / a);
        a += 1.0;
    }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0; t < 256; t++) {
        double x = bdf_normal(seed, sweep, BDF_P_GAMMA_N, entity, g, (int)(2 * t));
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = bdf_uniform(seed, sweep, BDF_P_GAMMA_U, entity, g, t);
        if (u < 1.0 - 0.0331 * (x * x) * (x * x)) return boost * d * v;
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
    }
    return boost * d;
}

// ---------------------------------------------------------------------------------------
// wave-level helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane)   // lane must be wave-uniform
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// kernel launch argument blocks ------------------------------------------------------------
#define BDF_K1_CODES 32            // distinct values up to which K1's coded variant keeps a per-wave table
struct TermDev {
    const int64_t *rowptr;
    const int32_t *colidx;
    const double *vals;
    const int32_t *perm;
    const double *linear;
    const double *fac[BDF_MAX_MODES - 1];
    int64_t nnz;
    int32_t n_other;
    int32_t lean;              // K1 lean gather: 1 = shared baseline, <= 2 other modes, factor matrices < 4 GiB with < 2^24
                               // rows (32-bit offsets); 2 = the same with 64-bit row offsets (D > 32 only); 0 = general path
    double alpha, mean;
    const uint32_t *packed;    // nullable: (value code << 24) | other-mode id per observation, with
    const double *table;       // the code -> value table (256 doubles)
    int32_t n_codes, _padc;
    const double *alpha_dev;   // nullable: the relation's precision in device memory (sampled on the device: sample_alpha inside bdf_gibbs_sweep); else `alpha`
};

__device__ __forceinline__ double term_alpha(const TermDev &T) { return T.alpha_dev ? *T.alpha_dev : T.alpha; }

struct SampleArgs {
    TermDev t[BDF_MAX_TERMS];
    int32_t n_terms, D;
    const double *mu;
    int32_t mu_is_matrix, _pad;
    const double *Lambda;
    uint32_t sweep, _pad3;
    uint64_t seed;
    uint32_t entity_tag, _pad2;
    double *out;
    const double *prior_b;     // Lambda mu (D) or Lambda mu_i (D x N), filled by the launch front-end
    const double *prior_c;     // index-reversed Lambda in the accumulator layout, filled by the launch front-end
    double *P_dump, *b_dump;
    int *flag;
    // nullable: the launch does not wait for the hyperprior draw that writes the prior pack; every wave polls *ready until
    // it reaches ready_want right before it adds the prior (bdf_gibbs_sweep on reserved CUs), and reads the pack past
    // the non-coherent caches
    const uint32_t *ready;
    uint32_t ready_want, _pad4;
    // nullable (bdf_ctx_span_next_rows): {start of the launch's first wave, end of its last} in s_memrealtime ticks (the 100 MHz
    // clock the XCDs share), by one atomic min / max per wave -- a launch's duration without events around it (k_rows_col only)
    unsigned long long *span;  // (64 shards of {start, end}: wave w uses shard w % 64)
    // nullable (bdf_gibbs_sweep, k_rows_col only): 64 counters, 16 words apart; a wave that has written its rows (write-through, drained)
    // adds 1 to counter (wave % 64): what the hyperprior chain polls instead of waiting for the launch's completion event
    uint32_t *done;
};
#define BDF_DONE_SHARDS 64
#define BDF_DONE_STRIDE 16         // words between two shards

int bdf_launch_sample_rows(bdf_ctx *ctx, const SampleArgs &a, const bdf_rel *const *rels, const int *modes, int shard,
                           int n_shards, bool dump);
int bdf_lr_launch(bdf_ctx *ctx, const SampleArgs &a, int64_t M_other, int64_t n_rows_entity, const void *items, int64_t n_items, int64_t n_padded, int64_t n32_padded,
                  const int32_t *rows_dev, bool transform, hipEvent_t e0, hipEvent_t e1);
int bdf_lr_max_observations();
int bdf_lr32_max_observations();
void bdf_plans_release(bdf_ctx *ctx, uint64_t rel_serial);

// ---- K1c (k_rows_col.hip): four rows per wave in the column layout ----------------------------------------------------
struct ColJob {           // one lane row of one round
    int32_t row;          // where the sample is written (the row's position in the factor matrix); -1: idle lane row
    int32_t orig;         // the row's ORIGINAL id: keys its random stream
    int64_t q_begin;      // first observation of the piece (index into the term's arrays)
    int32_t count;        // observations of the piece
    int32_t srow;         // a row that spans waves: its entry of the split-row table, else -1
    int32_t slot;         // ... and this part's slot in the slab
    int32_t flags;
};
#define COLF_LEADER 1     // the lane row that writes the sample of its group's row
#define COLF_PAIR 2       // the lane row's sums are added to its neighbour's (lane ^ 16)
#define COLF_QUAD 4       // ... and to the other half's (lane ^ 32)
#define COLF_MULTI 8      // the round is one part of a row that spans waves
struct ColSplit { int32_t slot_begin, n_slots; };
struct ColPlanDev {
    const ColJob *jobs;           // four per round, wave after wave
    const int32_t *wave_round;    // wave w runs rounds wave_round[w] .. wave_round[w + 1] - 1
    int32_t n_waves, _pad;
    const ColSplit *rows;
    double *partials;
    int32_t *arrived;             // per split row: parts that have published (self-resetting)
};
struct bdf_row_ref { int32_t out, orig; int64_t qb, cnt; };
struct bdf_col_plan {
    ColJob *jobs_dev = nullptr;
    int32_t *wave_round_dev = nullptr;
    ColSplit *rows_dev = nullptr;
    double *partials_dev = nullptr;
    int32_t *arrived_dev = nullptr;
    int32_t n_waves = 0, n_split_rows = 0;
    int64_t n_rounds = 0;
    double cost_max = 0.0, cost_min = 0.0;      // the planner's cost model: the heaviest and the lightest wave
};
int bdf_col_plan_build(bdf_ctx *ctx, const std::vector<bdf_row_ref> &rows, int T, int64_t slots, bdf_col_plan &plan);
void bdf_col_plan_free(bdf_col_plan &plan);
int bdf_col_launch(bdf_ctx *ctx, const SampleArgs &a, const bdf_col_plan &plan, int64_t M_other, hipEvent_t e0, hipEvent_t e1);
int bdf_predict_plain(bdf_ctx *ctx, const bdf_pairs *p, int D, const double *const *factors, double mean_value, double *out);   // rel_serial 0: every plan of the context
