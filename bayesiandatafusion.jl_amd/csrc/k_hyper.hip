// k_hyper.hip -- K6: hyperprior of an entity's latent rows.
//
//  bdf_hyper_sums   : N, sum_i U_i and U U' (src/sampling.jl:117-119) with U = sample - uhat (macau.jl:123),
//                     two-stage deterministic reduction.
//  bdf_hyper_sample : ConditionalNormalWishart (src/sampling.jl:116-127) + rand(::NormalWishart)
//                     (src/normal_wishart.jl:38-42) in ONE workgroup, so the D x D work never leaves the device.
//
// The Normal-Wishart draw, in the reference's terms:
//     W    = Tinv + UU' + b0 mu0 mu0' - beta_N mu_N mu_N'         (= inv(T_N))
//     Lam  = (L_T A)(L_T A)',  L_T = chol(T_N)' lower,  A = Bartlett matrix (A_aa = sqrt(chi2(nu_N - a)), A_ac ~ N(0,1), c < a)
//     mu   = mu_N + chol(inv(Lam) / beta_N)' z
// As in K1 both "Cholesky factor of an inverse" steps are obtained without forming the inverse: with W = U U'
// (U upper), L_T == U^-T; with Lam = U2 U2', chol(inv(Lam))' == U2^-T.  In index-reversed coordinates (~) these are
// ordinary lower Cholesky factors:  W~ = L~ L~',  Z~ = L~^-T (J A),  Lam~ = Z~ Z~',  Lam~ = L2~ L2~',
// mu~ = mu_N~ + L2~^-T z~ / sqrt(beta_N).
#include "bdf_common.h"
#include "wave_linalg.h"

namespace {

// ---- stage 1: per-block partial sums over a slice of rows ---------------------------------------------------
constexpr int HS_THREADS = 256;
constexpr int HS_TILE = 32;          // rows staged per iteration

__global__ __launch_bounds__(HS_THREADS) void k_hyper_partial(int D, int64_t N, const double *__restrict__ sample,
                                                               const double *__restrict__ uhat, int64_t rows_per_block,
                                                               double *__restrict__ partial)
{
    __shared__ double tile[HS_TILE][BDF_MAX_D + 1];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
    const int DD = D * D;
    constexpr int EPT = (BDF_MAX_D * BDF_MAX_D + HS_THREADS - 1) / HS_THREADS;
    double acc[EPT];
#pragma unroll
    for (int t = 0; t < EPT; t++) acc[t] = 0.0;
    double sacc = 0.0;
    for (int64_t base = r0; base < r1; base += HS_TILE) {
        const int nr = (int)((r1 - base < HS_TILE) ? (r1 - base) : HS_TILE);
        for (int idx = tid; idx < nr * D; idx += HS_THREADS) {
            const int rr = idx / D, e = idx % D;
            const int64_t off = (base + rr) * D + e;
            tile[rr][e] = sample[off] - (uhat ? uhat[off] : 0.0);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < EPT; t++) {
            const int e = tid + t * HS_THREADS;
            if (e < DD) {
                const int i = e % D, j = e / D;
                double s = acc[t];
                for (int rr = 0; rr < nr; rr++) s = fma(tile[rr][i], tile[rr][j], s);
                acc[t] = s;
            }
        }
        if (tid < D)
            for (int rr = 0; rr < nr; rr++) sacc += tile[rr][tid];
        __syncthreads();
    }
    double *p = partial + (int64_t)blockIdx.x * (DD + D);
#pragma unroll
    for (int t = 0; t < EPT; t++) {
        const int e = tid + t * HS_THREADS;
        if (e < DD) p[e] = acc[t];
    }
    if (tid < D) p[DD + tid] = sacc;
}

// ---- stage 2: fixed-order sum of the partials -----------------------------------------------------------------
// 16 lanes per output element: lane q sums partials q, q+16, ... and the 16 sums are combined by a butterfly -- a fixed
// order, so the result does not depend on scheduling
__global__ __launch_bounds__(256) void k_hyper_final(int D, int nblocks, const double *__restrict__ partial,
                                                     double *__restrict__ sumU, double *__restrict__ UUt)
{
    const int DD = D * D;
    const int q = threadIdx.x & 15;
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 4;
    double s = 0.0;
    if (e < DD + D)
        for (int b = q; b < nblocks; b += 16) s += partial[(int64_t)b * (DD + D) + e];
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (q == 0 && e < DD + D) {
        if (e < DD) UUt[e] = s;
        else sumU[e - DD] = s;
    }
}

// ---- Normal-Wishart draw on one wavefront -------------------------------------------------------------------
struct NWArgs {
    int D;
    double N;
    const double *sumU, *UUt, *mu0, *Tinv;
    double b0, nu;
    uint64_t seed;
    uint32_t sweep;
    uint32_t entity_tag;
    double *mu_out, *Lambda_out, *params_out;
    int *flag;
};

// One workgroup of 256 threads.  Wave 0 runs the two factorisations; all four waves draw the Bartlett matrix and form
// Lam~ = Z~ Z~'.  LDS images are row-major with leading dimension DP + 1.
template <int DP>
__global__ __launch_bounds__(256) void k_hyper_sample(NWArgs a)
{
    constexpr int LD = DP + 1;
    __shared__ double sA[DP * LD];      // Bartlett A~ = J A, then Z~
    __shared__ double sL[DP * LD];      // masked rows of Ah (factor of W~), later Lam~, later transposition image
    __shared__ double s_rp[64], s_sq[64], s_muN[64];
    __shared__ double s_tri[WL<DP>::TRI + 64];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D;
    const uint32_t sweep = a.sweep;
    const double beta_N = a.b0 + a.N;
    const double nu_N = a.nu + a.N;
    if (tid == 0) s_bad = 0;
    if (tid < 64) {
        const int e = D - 1 - tid;
        s_muN[tid] = (e >= 0) ? (a.b0 * a.mu0[e] + a.sumU[e]) / beta_N : 0.0;      // reversed: s_muN[c] = mu_N[D-1-c]
    }
    __syncthreads();

    if (wave == 0) {
        // ---- W~ = J W J, W = Tinv + UU' + b0 mu0 mu0' - beta_N mu_N mu_N' (Symmetric(): upper triangle), factorised
        const int c = lane % DP, ej = D - 1 - c;
        double col[DP];
#pragma unroll
        for (int i = 0; i < DP; i++) {
            const int ei = D - 1 - i;
            double w = (i == c) ? 1.0 : 0.0;
            if (ei >= 0 && ej >= 0) {
                const int lo = ei < ej ? ei : ej, hi = ei < ej ? ej : ei;
                w = a.Tinv[lo + (int64_t)hi * D] + a.UUt[lo + (int64_t)hi * D] + a.b0 * a.mu0[lo] * a.mu0[hi] -
                    beta_N * s_muN[D - 1 - lo] * s_muN[D - 1 - hi];
            }
            col[i] = w;
        }
        if (a.params_out && lane < DP && ej >= 0) {
            a.params_out[ej] = s_muN[c];
#pragma unroll
            for (int i = 0; i < DP; i++) {
                const int ei = D - 1 - i;
                if (ei >= 0) a.params_out[D + ei + (int64_t)ej * D] = col[i];
            }
        }
        double p_own, rp_own;
        if (wl_factor<DP, true>(col, p_own, rp_own, s_tri, lane) && lane == 0) s_bad = 1;
        if (lane < DP) {
#pragma unroll
            for (int k = 0; k < DP; k++) sL[c * LD + k] = col[k];         // Ah[c][k], k < c (else 0)
            s_rp[c] = rp_own;
            s_sq[c] = p_own * fast_rsqrt(p_own);
        }
    } else {
        // ---- Bartlett matrix, reversed rows: sA[i][c] = A[D-1-i][c];  A[r][c]: c < r normal, c == r sqrt(chi2(nu_N - r))
        for (int e = tid - 64; e < DP * DP; e += 192) {
            const int i = e / DP, c = e % DP;
            const int arow = D - 1 - i;
            double v = 0.0;
            if (arow >= 0 && c < D) {
                if (c < arow) v = bdf_normal(a.seed, sweep, BDF_P_NW_NORMAL, a.entity_tag, (uint64_t)arow, c);
                else if (c == arow) v = sqrt(2.0 * bdf_gamma(a.seed, sweep, a.entity_tag, (uint64_t)arow, 0.5 * (nu_N - (double)arow)));
            }
            sA[i * LD + c] = v;
        }
    }
    __syncthreads();

    // ---- Z~ = L~^-T A~  <=>  Ah' Z~ = diag(sqrt(p)) A~ : one thread per column, backward substitution
    if (tid < DP) {
        double z[DP];
#pragma unroll
        for (int i = DP - 1; i >= 0; i--) {
            double s = s_sq[i] * sA[i * LD + tid];
#pragma unroll
            for (int m = i + 1; m < DP; m++) s = fma(-sL[m * LD + i], z[m], s);
            z[i] = s * s_rp[i];
        }
#pragma unroll
        for (int i = 0; i < DP; i++) sA[i * LD + tid] = z[i];
    }
    __syncthreads();

    // ---- Lam~ = Z~ Z~' (identity on the padding), stored reversed in sL and natural in Lambda_out
    for (int e = tid; e < DP * DP; e += 256) {
        const int i = e / DP, j = e % DP;
        const int ei = D - 1 - i, ej = D - 1 - j;
        double s = 0.0;
        if (ei >= 0 && ej >= 0) {
            // fixed summation order in c; (i,j) and (j,i) multiply the same pairs: the result is exactly symmetric
            for (int c = 0; c < DP; c++) s = fma(sA[i * LD + c], sA[j * LD + c], s);
            a.Lambda_out[ei + (int64_t)ej * D] = s;
        } else {
            s = (i == j) ? 1.0 : 0.0;
        }
        sL[i * LD + j] = s;
    }
    __syncthreads();

    // ---- mu~ = mu_N~ + L2~^-T z~ / sqrt(beta_N), Lam~ = L2~ L2~'
    if (wave == 0) {
        const int c = lane % DP, ej = D - 1 - c;
        double col[DP];
#pragma unroll
        for (int i = 0; i < DP; i++) col[i] = sL[i * LD + c];
        wave_sync();
        double p_own, rp_own;
        if (wl_factor<DP, true>(col, p_own, rp_own, s_tri, lane) && lane == 0) s_bad = 1;
        double yh = 0.0;
        if (lane < DP && ej >= 0)
            yh = bdf_normal(a.seed, sweep, BDF_P_NW_MEAN, a.entity_tag, 0, ej) * (p_own * fast_rsqrt(p_own));
        const double x = wl_backward<DP, true>(s_tri, yh, rp_own, lane);
        if (lane < DP && ej >= 0) a.mu_out[ej] = s_muN[c] + x / sqrt(beta_N);
        wave_sync();
        if (lane == 0 && s_bad) atomicOr(a.flag, 2);
    }
}

}  // namespace

extern "C" int bdf_hyper_sums(bdf_ctx *ctx, int D, int64_t N, const double *sample, const double *uhat,
                              double *sumU, double *UUt)
{
    BDF_REQUIRE(ctx && sample && sumU && UUt, BDF_ERR_ARG, "bdf_hyper_sums: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_sums: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    BDF_REQUIRE(N >= 0, BDF_ERR_ARG, "bdf_hyper_sums: N < 0");
    int nblocks = (int)std::min<int64_t>(1024, (N + 31) / 32);
    if (nblocks < 1) nblocks = 1;
    int64_t rpb = (N + nblocks - 1) / nblocks;
    if (rpb < 1) rpb = 1;
    void *scratch;
    int rc = bdf_scratch(ctx, (size_t)nblocks * (D * D + D) * sizeof(double), &scratch);
    if (rc) return rc;
    hipLaunchKernelGGL(k_hyper_partial, dim3(nblocks), dim3(HS_THREADS), 0, ctx->stream, D, N, sample, uhat, rpb,
                       (double *)scratch);
    BDF_HIP(hipGetLastError());
    int tot = D * D + D;
    hipLaunchKernelGGL(k_hyper_final, dim3((tot + 15) / 16), dim3(256), 0, ctx->stream, D, nblocks,
                       (const double *)scratch, sumU, UUt);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

extern "C" int bdf_hyper_sample(bdf_ctx *ctx, int D, int64_t N, const double *sumU, const double *UUt,
                                const double *mu0, double b0, const double *Tinv, double nu, uint32_t entity_tag,
                                double *mu_out, double *Lambda_out, double *params_out)
{
    BDF_REQUIRE(ctx && sumU && UUt && mu0 && Tinv && mu_out && Lambda_out, BDF_ERR_ARG, "bdf_hyper_sample: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_sample: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    NWArgs a;
    a.D = D; a.N = (double)N; a.sumU = sumU; a.UUt = UUt; a.mu0 = mu0; a.Tinv = Tinv; a.b0 = b0; a.nu = nu;
    a.seed = ctx->seed; a.sweep = ctx->sweep_host; a.entity_tag = entity_tag;
    a.mu_out = mu_out; a.Lambda_out = Lambda_out; a.params_out = params_out; a.flag = ctx->flag_dev;
    if (D <= 16) hipLaunchKernelGGL(k_hyper_sample<16>, dim3(1), dim3(256), 0, ctx->stream, a);
    else if (D <= 32) hipLaunchKernelGGL(k_hyper_sample<32>, dim3(1), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_hyper_sample<64>, dim3(1), dim3(256), 0, ctx->stream, a);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}
