// k_sample_rows.hip -- K1: the latent-row sampler.
//
// Replaces sample_user_basic (src/sampling.jl:200-212 matrix, :215-234 tensor) and sample_user2
// (src/sampling.jl:266-289, sum over the entity's relations) of the reference, for every row of an
// entity at once (sample_latent_all2! :149-172, sample_user2_all! :251-264).
//
// Per row i (one 64-lane wavefront per row):
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
// Data flow: CSR of the relation in this mode (rowptr / other-mode ids / values, coalesced) -> gathered
// factor rows staged through LDS -> D x D accumulator in registers (lane = column) -> in-wave Cholesky
// (v_readlane broadcasts, no LDS) -> forward solve -> LDS transpose -> backward solve -> D doubles out.
#include "bdf_common.h"
#include "wave_linalg.h"

namespace {

constexpr int G = 16;   // observations staged per chunk

template <int DP>
struct Geo {
    static constexpr int NH = 64 / DP;        // lane groups per wave (DP=64:1, 32:2, 16:4)
    static constexpr int RPL = DP / NH;       // accumulator rows per lane (64, 16, 4)
};

template <int DP, bool DUMP>
__global__ __launch_bounds__(64) void k_sample_rows(SampleArgs a)
{
    constexpr int NH = Geo<DP>::NH, RPL = Geo<DP>::RPL;
    constexpr int STAGE = G * DP;
    constexpr int TBUF = DP * WL_TLD;
    __shared__ double smem[(STAGE + G > TBUF) ? (STAGE + G) : TBUF];
    double *srow = smem;            // [G][DP] staged w vectors (index-reversed)
    double *srr = smem + STAGE;     // [G] residuals y - base

    const int lane = threadIdx.x;
    const int c = lane % DP;        // column owned during accumulation (reversed coordinates)
    const int h = lane / DP;
    const int D = a.D;
    const int64_t row = a.rowlist ? (int64_t)a.rowlist[blockIdx.x] : (int64_t)blockIdx.x;
    const int ec = D - 1 - c;       // natural index of reversed column c (negative => padding)

    double tot[RPL];
#pragma unroll
    for (int t = 0; t < RPL; t++) tot[t] = 0.0;
    double btot = 0.0;

    for (int r = 0; r < a.n_terms; r++) {
        const TermDev &T = a.t[r];
        const int64_t beg = T.rowptr[row], end = T.rowptr[row + 1];
        double acc[RPL];
#pragma unroll
        for (int t = 0; t < RPL; t++) acc[t] = 0.0;
        double bacc = 0.0;
        for (int64_t q0 = beg; q0 < end; q0 += G) {
            const int g = (int)((end - q0 < G) ? (end - q0) : G);
            // ---- stage: lane (c,h) loads element ec of observations h, h+NH, ...
#pragma unroll
            for (int s = 0; s < G / NH; s++) {
                const int o = h + NH * s;
                double w = 0.0;
                if (o < g && ec >= 0) {
                    const int64_t q = q0 + o;
                    w = T.fac[0][(int64_t)T.colidx[q] * D + ec];
                    for (int k = 1; k < T.n_other; k++)
                        w *= T.fac[k][(int64_t)T.colidx[(int64_t)k * T.nnz + q] * D + ec];
                }
                srow[o * DP + c] = w;
            }
            if (lane < g) {
                const int64_t q = q0 + lane;
                const double base = T.linear ? T.linear[T.perm[q]] : T.mean;
                srr[lane] = T.vals[q] - base;
            }
            __syncthreads();
            // ---- rank-1 updates: acc[t] = S[h*RPL+t][c]
            for (int o = 0; o < g; o++) {
                const double vc = srow[o * DP + c];
                const double *vr = srow + o * DP + h * RPL;
#pragma unroll
                for (int t = 0; t < RPL; t++) acc[t] = fma(vr[t], vc, acc[t]);
                bacc = fma(vc, srr[o], bacc);
            }
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < RPL; t++) tot[t] = fma(T.alpha, acc[t], tot[t]);
        btot = fma(T.alpha, bacc, btot);
    }

    // ---- prior: P += Lambda, b += Lambda mu_i (reversed coordinates; identity padding)
    const double *mu_i = a.mu_is_matrix ? a.mu + row * D : a.mu;
#pragma unroll
    for (int t = 0; t < RPL; t++) {
        const int i = h * RPL + t;
        const int ei = D - 1 - i;
        double lam = 0.0;
        if (ei >= 0 && ec >= 0) lam = a.Lambda[ei + (int64_t)ec * D];
        else if (i == c) lam = 1.0;
        tot[t] += lam;
    }
    if (ec >= 0) {
        double s = 0.0;
        for (int j = 0; j < D; j++) s = fma(a.Lambda[ec + (int64_t)j * D], mu_i[j], s);
        btot += s;
    } else {
        btot = 0.0;
    }

    if (DUMP) {
        if (ec >= 0) {
#pragma unroll
            for (int t = 0; t < RPL; t++) {
                const int ei = D - 1 - (h * RPL + t);
                if (ei >= 0) a.P_dump[(row * D + ec) * D + ei] = tot[t];
            }
            if (h == 0) a.b_dump[row * D + ec] = btot;
        }
        return;
    }

    // ---- gather the full column c into lanes 0..DP-1
    double col[DP];
#pragma unroll
    for (int hh = 0; hh < NH; hh++)
#pragma unroll
        for (int t = 0; t < RPL; t++) col[hh * RPL + t] = __shfl(tot[t], c + hh * DP);
    double bj = __shfl(btot, c);     // lanes >= DP mirror lane c (harmless)

    // ---- in-wave Cholesky: afterwards lane j holds row j of L in col[0..j]
    double rinv_own;
    if (wl_chol_rows<DP>(col, rinv_own, lane) && lane == 0) atomicOr(a.flag, 1);

    // ---- forward solve L w = b
    bj = wl_fwd_rows<DP>(col, rinv_own, bj, lane);

    // ---- y = w + z  (z in reversed coordinates: lane j takes normal number D-1-j)
    const uint32_t sweep = *a.sweep;
    double yj = 0.0;
    if (lane < DP && ec >= 0) yj = bj + bdf_normal(a.seed, sweep, BDF_P_ROW, a.entity_tag, (uint64_t)row, ec);

    // ---- transpose L through LDS, then backward solve L' x = y
    wl_rows_to_cols<DP>(col, smem, lane);
    yj = wl_bwd_cols<DP>(col, rinv_own, yj, lane);

    if (lane < DP && ec >= 0) a.out[row * D + ec] = yj;
}

template <int DP>
int launch(bdf_ctx *ctx, const SampleArgs &a, bool dump)
{
    dim3 grid((unsigned)a.nrows), block(64);
    if (dump) hipLaunchKernelGGL((k_sample_rows<DP, true>), grid, block, 0, ctx->stream, a);
    else      hipLaunchKernelGGL((k_sample_rows<DP, false>), grid, block, 0, ctx->stream, a);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

}  // namespace

int bdf_launch_sample_rows(bdf_ctx *ctx, const SampleArgs &a, bool dump)
{
    if (a.nrows == 0) return BDF_OK;
    if (a.D <= 16) return launch<16>(ctx, a, dump);
    if (a.D <= 32) return launch<32>(ctx, a, dump);
    return launch<64>(ctx, a, dump);
}
