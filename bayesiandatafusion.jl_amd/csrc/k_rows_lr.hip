// k_rows_lr.hip -- K1-lr: the latent rows with FEW observations, sampled without a D x D factorisation.
//
// Replaces, for the rows it takes, sample_user_basic (src/sampling.jl:200-212) of the reference: the same conditional
// distribution  x_i ~ N(inv(P_i) b_i, inv(P_i)),  P_i = Lambda + alpha W W',  b_i = Lambda mu + alpha W r  (W = the D x n
// gathered factor rows of the row's n observations, r = values - mean), drawn by ANOTHER map from standard normals than the
// reference's  chol(inv(P_i))' z + inv(P_i) b_i.  A row with n << D observations has P_i = Lambda + (rank n): the reference's
// map costs D^2 n + D^3/3 per row whatever n; this one costs ~2 D n^2 + n^3/3 + 2 D^2 (the last term as a plain dense product
// over all rows).  With Lambda = L L' (lower Cholesky, once per launch) and Vt = V L^-T (the opposite entity's whole factor
// matrix transformed once per launch: wt_o = L^-1 w_o is then a gathered row of Vt):
//
//     e0  = L' mu + u                      u   = normals 0 .. D-1   of the row's stream (BDF_P_ROW, entity_tag, original id)
//     G   = I_n + alpha Wt' Wt             n x n, on the matrix cores
//     tau = G^-1 (r - Wt' e0 - delta / sqrt(alpha))      delta = normals D .. D+n-1
//     q   = e0 + alpha Wt tau
//     x   = L^-T q                         (k_rowmat: a dense N x D x D product over the rows of the launch, in place)
//
// z = 0 gives x = mu + alpha Lambda^-1 W G^-1 (r - W' mu), the posterior mean in its Kalman-gain form; the noise part is the
// N(0, (I + Phi' Phi)^-1) sampler of Bhattacharya, Chakraborty & Mallick (2016) with Phi = sqrt(alpha) Wt'.  The oracle holds
// the same function (oracle/bdf_oracle.c: orc_sample_row_lowrank) and tests/test_oracle_known_answers.py proves -- the map
// is affine in z -- that its mean is inv(P_i) b_i and its S S' is inv(P_i) to 1e-10 for every n in 0 .. D/2 + 1.
// Parity: this kernel against that function on the same normals (1e-8), and >= 10^5-draw moments on the device.
//
// One wavefront per row (n <= 15).  The n rows of Vt arrive with the row kernel's coalesced gather (lane (j, h): element
// 16 I + j of observation 4 k + h), go to LDS once and come back in the MFMA operand layout with the contraction over D
// (lane (i, kk): elements kk D/4 .. of observation i); e0 rides along as "observation 15", which makes Wt' e0 column 15 of
// the same sixteen v_mfma_f64_16x16x4_f64.  G is then factored in the accumulator layout by the row kernel's own 16 x 16
// routines (c_layout_chol.h), the right-hand side riding along as the extra row.
#include "bdf_common.h"
#include "wave_linalg.h"
#include "c_layout_chol.h"

namespace {

template <int DP>
struct LrGeo {
    static constexpr int DB = DP / 16;
    static constexpr int KQ = DP / 4;                  // contraction elements per lane row of the MFMA operand
    static constexpr int LD = DP + 2;                  // doubles between staged rows: even (16-byte reads), odd multiple of two banks
    static constexpr int STAGE = 16 * LD;
    static constexpr int TRI = Geo<16>::WAVE_LDS;      // packed 16 x 16 factor (+ the extra row's panel); the normals sit there first
    static constexpr int WAVE_LDS = STAGE + TRI;
    static_assert(TRI >= DP + 16, "the D + n normals of a row fit the packed factor's space");
};

// ---- the launch's constants: L = chol(Lambda) (lower, natural order), then
//      Tf[d][c] = L^-T[d][c] = L^-1[c][d]   (Vt = V Tf: row m of Vt is L^-1 v_m)
//      Tb[d][c] = L^-1[d][c]                (x' = q' Tb:  x = L^-T q)
//      mt[d]    = (L' mu)[d]
// both matrices DP x DP row-major, zero outside D x D.  One wavefront; the matrix lives in LDS.
template <int DP>
__global__ __launch_bounds__(64) void k_lr_prep(int D, const double *__restrict__ Lambda, const double *__restrict__ mu,
                                                 double *__restrict__ Tf, double *__restrict__ Tb, double *__restrict__ mt, int *flag)
{
    constexpr int LDL = DP + 1;
    // sA[i * LDL + c]: the Schur complement's lower triangle, then L in place; L^-1 (lower triangular too) goes TRANSPOSED
    // into the strict upper triangle -- X[i][c], i > c, at sA[c * LDL + i] -- and its diagonal into sD
    __shared__ double sA[DP * LDL];
    __shared__ double sD[DP];
    const int c = threadIdx.x;
    for (int e = c; e < DP * DP; e += 64) {
        const int i = e / DP, cc = e % DP;
        sA[i * LDL + cc] = (i < D && cc < D) ? Lambda[i + (int64_t)cc * D] : ((i == cc) ? 1.0 : 0.0);
    }
    wave_sync();
    bool bad = false;
    for (int k = 0; k < D; k++) {
        const double p = sA[k * LDL + k];
        if (!(p > 0.0)) bad = true;
        const double sd = sqrt(p);
        const double lck = (c < D && c > k) ? sA[c * LDL + k] / sd : 0.0;        // L[c][k]
        wave_sync();
        if (c == k) sA[k * LDL + k] = sd;
        else if (c < D && c > k) sA[c * LDL + k] = lck;
        wave_sync();
        // lane c updates ROW c of the trailing lower triangle: A[c][m] -= L[c][k] L[m][k], k < m <= c
        if (c < D && c > k)
            for (int m = k + 1; m <= c; m++) sA[c * LDL + m] = fma(-lck, sA[m * LDL + k], sA[c * LDL + m]);
        wave_sync();
    }
    if (bad && c == 0) atomicOr(flag, 1);
    // L^-1 by columns: lane c solves L x = e_c (x_i = 0 above the diagonal); its entries sit in row c of the upper triangle
    if (c < D) {
        const double xd = 1.0 / sA[c * LDL + c];
        sD[c] = xd;
        for (int i = c + 1; i < D; i++) {
            double s = -sA[i * LDL + c] * xd;
            for (int m = c + 1; m < i; m++) s = fma(-sA[i * LDL + m], sA[c * LDL + m], s);
            sA[c * LDL + i] = s / sA[i * LDL + i];
        }
    }
    wave_sync();
    for (int e = c; e < DP * DP; e += 64) {
        const int d = e / DP, cc = e % DP;
        const bool in = d < D && cc < D;
        // X[d][cc] (d >= cc) and X[cc][d] (cc >= d)
        const double lo = (d > cc) ? sA[cc * LDL + d] : ((d == cc && in) ? sD[d] : 0.0);
        const double up = (cc > d) ? sA[d * LDL + cc] : ((d == cc && in) ? sD[d] : 0.0);
        Tb[e] = in ? lo : 0.0;
        Tf[e] = in ? up : 0.0;
    }
    if (c < DP) {
        double s = 0.0;
        if (c < D)
            for (int i = c; i < D; i++) s = fma(sA[i * LDL + c], mu[i], s);
        mt[c] = s;
    }
}

// ---- Y = X T for rows of D doubles (X, Y row-major with leading dimension D; T: DP x DP row-major, zero-padded), on the
// matrix cores.  A tile is 16 rows; the DB = DP / 16 waves of a tile each take one 16-column block of the result and keep
// their block of T in registers; the contraction runs over d = kk DP/4 + s (lane row kk, k-step s), so that a lane reads
// DP/4 consecutive doubles of its row.  rows == nullptr: rows 0 .. n_rows-1; else the listed rows (entries < 0: none).
// In place (Y == X) is allowed: every wave of a tile has read the tile before any of them writes (workgroup barrier).
template <int DP>
__global__ __launch_bounds__(256) void k_rowmat(const double *X, double *Y, const double *__restrict__ T, int D, const int32_t *__restrict__ rows,
                                                int64_t n_rows, int64_t n_iters)
{
    constexpr int DB = DP / 16, KQ = DP / 4, TPW = 4 / DB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int sub = wave / DB, cb = wave % DB;
    double b[KQ];
#pragma unroll
    for (int s = 0; s < KQ; s++) b[s] = T[(kk * KQ + s) * DP + 16 * cb + i];
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t r = (it * TPW + sub) * 16 + i;
        int64_t row = -1;
        if (r < n_rows) row = rows ? (int64_t)rows[r] : r;
        double a[KQ];
        if (row >= 0) {
            const double *src = X + row * D + kk * KQ;
            if (D == DP) {
#pragma unroll
                for (int s = 0; s < KQ; s += 2) { const d2 v = *(const d2 *)(src + s); a[s] = v[0]; a[s + 1] = v[1]; }
            } else {
#pragma unroll
                for (int s = 0; s < KQ; s++) a[s] = (kk * KQ + s < D) ? src[s] : 0.0;
            }
        } else {
#pragma unroll
            for (int s = 0; s < KQ; s++) a[s] = 0.0;
        }
        __syncthreads();
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KQ; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
        // C layout: lane (j = i, h = kk), register r: tile row h + 4 r, column 16 cb + j
        const int col = 16 * cb + i;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int64_t rowm = __shfl((long long)row, kk + 4 * rr);
            if (rowm >= 0 && col < D) Y[rowm * D + col] = acc[rr];
        }
    }
}

struct LrItem {
    int32_t row;          // where the sample is written (position in the factor matrix)
    int32_t orig;         // the row's original id (random stream)
    int64_t q_begin;
    int32_t count, _pad;
};

struct LrArgs {
    const int32_t *colidx;      // the relation's other-mode ids, mode order
    const double *vals;
    const double *vt;           // the opposite factor transformed: rows of D doubles
    const double *mt;           // L' mu
    double *out;
    double alpha, mean;
    uint64_t seed;
    uint32_t sweep, entity_tag;
    int32_t D, _pad;
    int *flag;
};

template <int DP>
__global__ __launch_bounds__(256, (DP == 64 ? 4 : 5)) void k_rows_lr(LrArgs a, const LrItem *__restrict__ items, int64_t n_items)
{
    using LG = LrGeo<DP>;
    using G16 = Geo<16>;
    constexpr int DB = LG::DB, KQ = LG::KQ, LD = LG::LD;
    __shared__ __attribute__((aligned(16))) double lds[4 * LG::WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    if (w >= n_items) return;
    double *st = lds + wave * LG::WAVE_LDS, *tri = st + LG::STAGE;
    const int j = lane & 15, h = lane >> 4;
    const LrItem it = items[w];
    const int D = a.D, n = it.count;

    // ---- the row's D + n normals, one Philox block and one Box-Muller pair per lane, through LDS (the packed factor's space)
    if (2 * lane < D + n) {
        double z0, z1;
        bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, (uint32_t)lane, z0, z1);
        tri[2 * lane] = z0;
        tri[2 * lane + 1] = z1;
    }
    // ---- gather: lane (j, h) takes element 16 I + j of observations h, h + 4, h + 8, h + 12
    double wv[4][DB];
    double rv = 0.0;
    if (n > 0) {
        int64_t ix[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int o = 4 * k + h;
            ix[k] = a.colidx[it.q_begin + (o < n ? o : n - 1)];
        }
        if (j < n) rv = a.vals[it.q_begin + j] - a.mean;
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int I = 0; I < DB; I++) {
                const int e = 16 * I + j;
                const double v = a.vt[ix[k] * D + (e < D ? e : 0)];
                wv[k][I] = (4 * k + h < n && e < D) ? v : 0.0;
            }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int I = 0; I < DB; I++) wv[k][I] = 0.0;
    }
    wave_sync();
    // e0 = L' mu + u in the lanes of lane row 3 (element 16 I + j), staged as "observation 15"; delta_j in lane j
    double e0[DB];
#pragma unroll
    for (int I = 0; I < DB; I++) {
        const int e = 16 * I + j;
        e0[I] = (e < D) ? a.mt[e] + tri[e] : 0.0;
    }
    const double dl = (j < n) ? tri[D + j] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int I = 0; I < DB; I++)
            st[(4 * k + h) * LD + 16 * I + j] = (k == 3 && h == 3) ? e0[I] : wv[k][I];
    wave_sync();
    // ---- G~ = S S' with S = [Wt' ; 0 ; e0'] (16 x D): operand lane (i = j, kk = h), k-step s: S[i][kk KQ + s]
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    {
        const double *src = st + j * LD + h * KQ;
        double x[KQ];
#pragma unroll
        for (int s = 0; s < KQ; s += 2) { const d2 v = *(const d2 *)(src + s); x[s] = v[0]; x[s + 1] = v[1]; }
#pragma unroll
        for (int s = 0; s < KQ; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s], x[s], acc, 0, 0, 0);
    }
    // ---- the n x n system G tau = rho in the accumulator layout (lane (j, h), register r: row h + 4 r, column j),
    // identity outside n x n; Wt' e0 is row 15 of the product
    const double g = __shfl(acc[3], 48 + j);
    double A[4], bv[1], ts[1];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int m = h + 4 * r;
        const double id = (m == j) ? 1.0 : 0.0;
        A[r] = (m < n && j < n) ? fma(a.alpha, acc[r], id) : id;
    }
    bv[0] = (j < n) ? rv - g - dl * fast_rsqrt(a.alpha) : 0.0;
    ts[0] = 0.0;
    const int Ds = n > 1 ? n : 1;
    wave_sync();                                       // the normals have been read: the space is the packed factor's now
    zero_packed_factor<16>(tri, lane);
    factor_all_blocked<16>(A, bv, ts, tri, j, h, Ds, std::make_integer_sequence<int, 15>{});
    const typename G16::ColRT cr = G16::col_rt(lane < 16 ? lane : 0);
    wave_sync();
    double dv = 1.0, tv = 0.0;
    if (lane < Ds) { dv = tri[cr.cbase + (lane & 3) * cr.nr4]; tv = ts[0]; }
    if (!(dv > 0.0)) atomicOr(a.flag, 1);
    const double rdv = fast_rcp(dv);
    double yh = tv;
    unsigned colq[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        colq[q] = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(tri + cr.cbase + q * cr.nr4 - cr.q);
    backward_all<16>(yh, rdv, colq, std::make_integer_sequence<int, 1>{});
    const double tau = (lane < n) ? yh * rdv : 0.0;
    // ---- q = e0 + alpha Wt tau: lane (j, h) adds its observations' parts, then the four lane rows are summed
    double t4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) t4[k] = __shfl(tau, 4 * k + h);
    double qv = 0.0;
#pragma unroll
    for (int I = 0; I < DB; I++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) s = fma(wv[k][I], t4[k], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const double e = st[15 * LD + 16 * I + j];
        const double v = fma(a.alpha, s, e);
        qv = (h == I) ? v : qv;
    }
    const int e = 16 * h + j;
    if (h < DB && e < D) a.out[(int64_t)it.row * D + e] = qv;
}

int lr_buffers(bdf_ctx *ctx, size_t vt_bytes)
{
    if (!ctx->lr_T) BDF_HIP(hipMalloc((void **)&ctx->lr_T, (size_t)(2 * 64 * 64 + 64) * sizeof(double)));
    if (vt_bytes > ctx->lr_vt_bytes) {
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->lr_vt) BDF_HIP(hipFree(ctx->lr_vt));
        ctx->lr_vt = nullptr; ctx->lr_vt_bytes = 0;
        const size_t nb = (vt_bytes + 255) & ~(size_t)255;
        BDF_HIP(hipMalloc((void **)&ctx->lr_vt, nb));
        ctx->lr_vt_bytes = nb;
    }
    return BDF_OK;
}

template <int DP>
int lr_launch_t(bdf_ctx *ctx, const SampleArgs &a, int64_t M_other, const void *items, int64_t n_items, const int32_t *rows_dev, bool transform,
                hipEvent_t e0, hipEvent_t e1)
{
    const int D = a.D;
    double *Tf = ctx->lr_T, *Tb = Tf + 64 * 64, *mt = Tb + 64 * 64;
    constexpr int TPW = 4 / (DP / 16);
    if (transform) {
        hipExtLaunchKernelGGL((k_lr_prep<DP>), dim3(1), dim3(64), 0, ctx->stream, e0, nullptr, 0, D, a.Lambda, a.mu, Tf, Tb, mt, a.flag);
        e0 = nullptr;
        const int64_t iters = (M_other + 16 * TPW - 1) / (16 * TPW);
        hipLaunchKernelGGL((k_rowmat<DP>), dim3((unsigned)std::min<int64_t>(iters, 4096)), dim3(256), 0, ctx->stream, a.t[0].fac[0], ctx->lr_vt,
                           (const double *)Tf, D, (const int32_t *)nullptr, M_other, iters);
    }
    LrArgs la;
    la.colidx = a.t[0].colidx; la.vals = a.t[0].vals; la.vt = ctx->lr_vt; la.mt = mt; la.out = a.out;
    la.alpha = a.t[0].alpha; la.mean = a.t[0].mean; la.seed = a.seed; la.sweep = a.sweep; la.entity_tag = a.entity_tag;
    la.D = D; la._pad = 0; la.flag = a.flag;
    hipExtLaunchKernelGGL((k_rows_lr<DP>), dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, ctx->stream, e0, nullptr, 0, la, (const LrItem *)items, n_items);
    const int64_t iters = (n_items + 16 * TPW - 1) / (16 * TPW);
    hipExtLaunchKernelGGL((k_rowmat<DP>), dim3((unsigned)std::min<int64_t>(iters, 4096)), dim3(256), 0, ctx->stream, nullptr, e1, 0, (const double *)a.out, a.out,
                          (const double *)Tb, D, rows_dev, n_items, iters);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

}  // namespace

// The rows `items` (LrItem: one two-mode relation, at most 15 observations each, shared prior mean) of the launch described by
// `a`.  transform: L = chol(Lambda), the opposite factor's M_other rows transformed into the context's buffer (false: both
// are still valid from the previous chunk of the same entity launch).  rows_dev: the rows' positions (n_items int32).
int bdf_lr_launch(bdf_ctx *ctx, const SampleArgs &a, int64_t M_other, const void *items, int64_t n_items, const int32_t *rows_dev, bool transform,
                  hipEvent_t e0, hipEvent_t e1)
{
    int rc = lr_buffers(ctx, (size_t)M_other * a.D * sizeof(double));
    if (rc) return rc;
    if (a.D <= 32) return lr_launch_t<32>(ctx, a, M_other, items, n_items, rows_dev, transform, e0, e1);
    return lr_launch_t<64>(ctx, a, M_other, items, n_items, rows_dev, transform, e0, e1);
}
