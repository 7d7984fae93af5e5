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
#include <algorithm>

namespace {

// ---- stage 1: per-block partial sums over a slice of rows -------------------------------------------------------
// U U' = sum over rows of u u' is the same rank-4 MFMA update as K1's (k_sample_rows.hip): lane (j = l & 15, h = l >> 4)
// supplies element 16 I + j of row 4 s + h, straight from global memory (a coalesced 128-byte read per 16 lanes), and the
// lower block-triangle accumulates in the MFMA C layout.  The four waves of a workgroup take every fourth 4-row step of the
// block's slice, all of a wave's loads are issued before its first MFMA, and the waves' results are added in wave order.
typedef double hd4 __attribute__((ext_vector_type(4)));
constexpr int HS_THREADS = 256;
constexpr int HS_ROWS = 128;         // rows per workgroup: 8 steps of 4 rows per wave

template <int DP>
struct HGeo {
    static constexpr int DB = DP / 16, NB = DB * (DB + 1) / 2;
    static constexpr int PSZ = NB * 4 * 64 + DB * 16;      // doubles per partial: C-layout blocks, then the column sums
};

template <int DP>
__global__ __launch_bounds__(HS_THREADS) void k_hyper_partial(int D, int64_t N, const double *__restrict__ sample,
                                                               const double *__restrict__ uhat, double *__restrict__ partial)
{
    constexpr int DB = HGeo<DP>::DB, NB = HGeo<DP>::NB, PSZ = HGeo<DP>::PSZ;
    constexpr int KS = HS_ROWS / 16;                     // steps per wave
    __shared__ double red[3 * PSZ];
    __builtin_amdgcn_s_setprio(3);      // small and on the sweep's critical path, usually beside a chip-filling K1 launch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, h = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * HS_ROWS;
    double u[KS][DB];
#pragma unroll
    for (int k = 0; k < KS; k++) {
        const int64_t row = r0 + 4 * (wave + 4 * k) + h;
#pragma unroll
        for (int I = 0; I < DB; I++) {
            const int e = 16 * I + j;
            const bool ok = row < N && e < D;
            const int64_t off = (ok ? row : 0) * D + (ok ? e : 0);
            const double v = sample[off] - (uhat ? uhat[off] : 0.0);
            u[k][I] = ok ? v : 0.0;
        }
    }
    hd4 acc[NB];
    double cs[DB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = hd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < DB; I++) cs[I] = 0.0;
#pragma unroll
    for (int k = 0; k < KS; k++) {
        int b = 0;
#pragma unroll
        for (int I = 0; I < DB; I++) {
#pragma unroll
            for (int J = 0; J <= I; J++) {
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[k][I], u[k][J], acc[b], 0, 0, 0);
                b++;
            }
            cs[I] += u[k][I];
        }
    }
#pragma unroll
    for (int I = 0; I < DB; I++) {
        cs[I] += __shfl_xor(cs[I], 16);
        cs[I] += __shfl_xor(cs[I], 32);
    }
    if (wave > 0) {
        double *dst = red + (wave - 1) * PSZ;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(b * 4 + r) * 64 + lane] = acc[b][r];
        if (lane < 16)
#pragma unroll
            for (int I = 0; I < DB; I++) dst[NB * 4 * 64 + I * 16 + lane] = cs[I];
    }
    __syncthreads();
    if (wave == 0) {
        double *p = partial + (int64_t)blockIdx.x * PSZ;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double v = acc[b][r];
#pragma unroll
                for (int w = 0; w < 3; w++) v += red[w * PSZ + (b * 4 + r) * 64 + lane];
                p[(b * 4 + r) * 64 + lane] = v;
            }
        if (lane < 16)
#pragma unroll
            for (int I = 0; I < DB; I++) {
                double v = cs[I];
#pragma unroll
                for (int w = 0; w < 3; w++) v += red[w * PSZ + NB * 4 * 64 + I * 16 + lane];
                p[NB * 4 * 64 + I * 16 + lane] = v;
            }
    }
}

// ---- stage 2: fixed-order sum of the partials -----------------------------------------------------------------
// 16 lanes per partial element: lane q sums blocks q, q+16, ... and the 16 sums are combined by a butterfly -- a fixed
// order, so the result does not depend on scheduling.  The C-layout element (block (I,J), register r, lane l) is entry
// (16 I + (l >> 4) + 4 r, 16 J + (l & 15)) of U U' and, for an off-diagonal block, its mirror image.
template <int DP>
__global__ __launch_bounds__(256) void k_hyper_final(int D, int nblocks, const double *__restrict__ partial,
                                                     double *__restrict__ sumU, double *__restrict__ UUt)
{
    constexpr int DB = HGeo<DP>::DB, NB = HGeo<DP>::NB, PSZ = HGeo<DP>::PSZ;
    __builtin_amdgcn_s_setprio(3);
    const int q = threadIdx.x & 15;
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 4;
    double s = 0.0;
    if (e < PSZ)
        for (int b = q; b < nblocks; b += 16) s += partial[(int64_t)b * PSZ + e];
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (q != 0 || e >= PSZ) return;
    if (e < NB * 4 * 64) {
        const int b = e >> 8, r = (e >> 6) & 3, l = e & 63;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= b) I++;
        const int J = b - I * (I + 1) / 2;
        const int row = 16 * I + (l >> 4) + 4 * r, col = 16 * J + (l & 15);
        if (row < D && col < D) {
            UUt[row + (int64_t)col * D] = s;
            if (I != J) UUt[col + (int64_t)row * D] = s;
        }
    } else {
        const int el = e - NB * 4 * 64;                   // 16 I + j
        if (el < D) sumU[el] = s;
    }
    (void)DB;
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
    const double *draws;       // nullable: Bartlett matrix + mean normals from k_hyper_draws (bdf_hyper_draws)
    double *pack_out;          // nullable: Lambda mu (D) then the accumulator-layout image of the reversed Lambda (K1)
    int *flag;
};

// The random part of the draw does not depend on the data: the Bartlett matrix A (A_aa = sqrt(chi2(nu_N - a)),
// A_ac ~ N(0,1) for c < a, row-major D x D) and the D normals of the mean can be drawn while the rows are still being
// sampled (bdf_hyper_draws, one lane per entry over many workgroups), which takes the slowest part of k_hyper_sample --
// the gamma rejection loops -- off the sweep's critical path.  Same streams and values as the in-kernel draw.
__global__ __launch_bounds__(64) void k_hyper_draws(int D, double nu_N, uint64_t seed, uint32_t sweep, uint32_t entity_tag,
                                                    double *out)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e < D * D) {
        const int arow = e / D, c = e % D;
        double v = 0.0;
        if (c < arow) v = bdf_normal(seed, sweep, BDF_P_NW_NORMAL, entity_tag, (uint64_t)arow, c);
        else if (c == arow) v = sqrt(2.0 * bdf_gamma(seed, sweep, entity_tag, (uint64_t)arow, 0.5 * (nu_N - (double)arow)));
        out[e] = v;
    } else if (e < D * D + D) {
        out[e] = bdf_normal(seed, sweep, BDF_P_NW_MEAN, entity_tag, 0, e - D * D);
    }
}

// One workgroup of 256 threads.  Wave 0 runs the two factorisations; all four waves draw the Bartlett matrix and form
// Lam~ = Z~ Z~'.  LDS images are row-major with leading dimension DP + 1.
#ifdef BDF_HYPER_STAMPS
#define HSTAMP(k) do { if (threadIdx.x == 0 && a.params_out) ((unsigned long long *)a.params_out)[a.D + a.D * a.D + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HSTAMP(k) do { } while (0)
#endif

template <int DP>
__global__ __launch_bounds__(256) void k_hyper_sample(NWArgs a)
{
    constexpr int LD = DP + 1;
    __builtin_amdgcn_s_setprio(3);      // one workgroup beside a chip-filling K1 launch: take the issue slots when ready
    __shared__ double sA[DP * LD];      // Bartlett A~ = J A, then Z~
    __shared__ double sL[DP * LD];      // masked rows of Ah (factor of W~), later Lam~, later transposition image
    __shared__ double s_rp[64], s_sq[64], s_muN[64], s_mu[64];
    __shared__ double s_tri[WL<DP>::TRI + 64];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D;
    const uint32_t sweep = a.sweep;
    const double beta_N = a.b0 + a.N;
    const double nu_N = a.nu + a.N;
    HSTAMP(0);
    if (tid == 0) s_bad = 0;
    if (tid < 64) {
        const int e = D - 1 - tid;
        s_muN[tid] = (e >= 0) ? (a.b0 * a.mu0[e] + a.sumU[e]) / beta_N : 0.0;      // reversed: s_muN[c] = mu_N[D-1-c]
    }
    __syncthreads();

    HSTAMP(1);
    // ---- W~ = J W J, W = Tinv + UU' + b0 mu0 mu0' - beta_N mu_N mu_N' (Symmetric(): upper triangle): all threads, into sL
    for (int e = tid; e < DP * DP; e += 256) {
        const int i = e / DP, c = e % DP;
        const int ei = D - 1 - i, ej = D - 1 - c;
        double w = (i == c) ? 1.0 : 0.0;
        if (ei >= 0 && ej >= 0) {
            const int lo = ei < ej ? ei : ej, hi = ei < ej ? ej : ei;
            w = a.Tinv[lo + (int64_t)hi * D] + a.UUt[lo + (int64_t)hi * D] + a.b0 * a.mu0[lo] * a.mu0[hi] -
                beta_N * s_muN[D - 1 - lo] * s_muN[D - 1 - hi];
            if (a.params_out) a.params_out[D + ei + (int64_t)ej * D] = w;
        }
        sL[i * LD + c] = w;
    }
    if (a.params_out && tid < DP && D - 1 - tid >= 0) a.params_out[D - 1 - tid] = s_muN[tid];
    __syncthreads();
    if (wave == 0) {
        const int c = lane % DP;
        double col[DP];
#pragma unroll
        for (int i = 0; i < DP; i++) col[i] = sL[i * LD + c];
        wave_sync();
        double p_own, rp_own;
        HSTAMP(2);
        if (wl_factor<DP, true>(col, p_own, rp_own, s_tri, lane) && lane == 0) s_bad = 1;
        HSTAMP(3);
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
                if (a.draws) v = a.draws[arow * D + c];
                else if (c < arow) v = bdf_normal(a.seed, sweep, BDF_P_NW_NORMAL, a.entity_tag, (uint64_t)arow, c);
                else if (c == arow) v = sqrt(2.0 * bdf_gamma(a.seed, sweep, a.entity_tag, (uint64_t)arow, 0.5 * (nu_N - (double)arow)));
            }
            sA[i * LD + c] = v;
        }
    }
    __syncthreads();

    HSTAMP(4);
    // ---- Z~ = L~^-T A~  <=>  Ah' Z~ = diag(sqrt(p)) A~ : one thread per column, backward substitution
    if (tid < DP) {
        double z[DP];
#pragma unroll
        for (int i = DP - 1; i >= 0; i--) {
            // four interleaved partial sums (fixed assignment m % 4): four short dependency chains instead of one long one
            double s4[4] = {s_sq[i] * sA[i * LD + tid], 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = i + 1; m < DP; m++) s4[m & 3] = fma(-sL[m * LD + i], z[m], s4[m & 3]);
            z[i] = ((s4[0] + s4[1]) + (s4[2] + s4[3])) * s_rp[i];
        }
#pragma unroll
        for (int i = 0; i < DP; i++) sA[i * LD + tid] = z[i];
    }
    __syncthreads();

    HSTAMP(5);
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

    HSTAMP(6);
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
            yh = (a.draws ? a.draws[D * D + ej] : bdf_normal(a.seed, sweep, BDF_P_NW_MEAN, a.entity_tag, 0, ej)) *
                 (p_own * fast_rsqrt(p_own));
        const double x = wl_backward<DP, true>(s_tri, yh, rp_own, lane);
        const double mu_c = s_muN[c] + x / sqrt(beta_N);
        if (lane < DP && ej >= 0) a.mu_out[ej] = mu_c;
        if (lane < DP) s_mu[c] = (ej >= 0) ? mu_c : 0.0;          // reversed: s_mu[c] = mu[D-1-c]
        wave_sync();
        if (lane == 0 && s_bad) atomicOr(a.flag, 2);
        HSTAMP(7);
    }
    if (a.pack_out == nullptr) return;
    __syncthreads();
    // ---- what the row sampler needs of (mu, Lambda), written here so that it needs no pre-launch of its own:
    // Lambda mu (same products in the same order as k_prior of k_sample_rows.hip: bit-identical), and Lam~ in
    // the MFMA accumulator layout [block * 4 + r][lane] (identity on the padding -- exactly what sL holds)
    for (int e = tid >> 3; e < D; e += 32) {                      // eight lanes per entry: lane part p adds i = p, p+8, ...
        const int part = tid & 7;
        double v = 0.0;
        for (int i = part; i < D; i += 8) v = fma(sL[(D - 1 - e) * LD + (D - 1 - i)], s_mu[D - 1 - i], v);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (part == 0) a.pack_out[e] = v;
    }
    constexpr int DB = DP / 16;
    for (int e = wave; e < DB * (DB + 1) / 2 * 4; e += 4) {
        const int b = e >> 2, r = e & 3;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= b) I++;
        const int J = b - I * (I + 1) / 2;
        a.pack_out[D + e * 64 + lane] = sL[(16 * I + (lane >> 4) + 4 * r) * LD + 16 * J + (lane & 15)];
    }
    HSTAMP(8);
}

}  // namespace

extern "C" int bdf_hyper_sums(bdf_ctx *ctx, int D, int64_t N, const double *sample, const double *uhat,
                              double *sumU, double *UUt)
{
    BDF_REQUIRE(ctx && sample && sumU && UUt, BDF_ERR_ARG, "bdf_hyper_sums: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_sums: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    BDF_REQUIRE(N >= 0, BDF_ERR_ARG, "bdf_hyper_sums: N < 0");
    const int nblocks = (int)std::max<int64_t>(1, (N + HS_ROWS - 1) / HS_ROWS);
    const int DP = D <= 16 ? 16 : (D <= 32 ? 32 : 64);
    const int psz = DP == 16 ? HGeo<16>::PSZ : (DP == 32 ? HGeo<32>::PSZ : HGeo<64>::PSZ);
    void *scratch;
    int rc = bdf_scratch(ctx, (size_t)nblocks * psz * sizeof(double), &scratch);
    if (rc) return rc;
    double *part = (double *)scratch;
    const dim3 fgrid((psz + 15) / 16);
    if (DP == 16) {
        hipLaunchKernelGGL(k_hyper_partial<16>, dim3(nblocks), dim3(HS_THREADS), 0, ctx->stream, D, N, sample, uhat, part);
        hipLaunchKernelGGL(k_hyper_final<16>, fgrid, dim3(256), 0, ctx->stream, D, nblocks, (const double *)part, sumU, UUt);
    } else if (DP == 32) {
        hipLaunchKernelGGL(k_hyper_partial<32>, dim3(nblocks), dim3(HS_THREADS), 0, ctx->stream, D, N, sample, uhat, part);
        hipLaunchKernelGGL(k_hyper_final<32>, fgrid, dim3(256), 0, ctx->stream, D, nblocks, (const double *)part, sumU, UUt);
    } else {
        hipLaunchKernelGGL(k_hyper_partial<64>, dim3(nblocks), dim3(HS_THREADS), 0, ctx->stream, D, N, sample, uhat, part);
        hipLaunchKernelGGL(k_hyper_final<64>, fgrid, dim3(256), 0, ctx->stream, D, nblocks, (const double *)part, sumU, UUt);
    }
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

extern "C" int bdf_hyper_draws(bdf_ctx *ctx, int D, int64_t N, double nu, uint32_t entity_tag, double *draws_out)
{
    BDF_REQUIRE(ctx && draws_out, BDF_ERR_ARG, "bdf_hyper_draws: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_draws: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    const int total = D * D + D;
    hipLaunchKernelGGL(k_hyper_draws, dim3((total + 63) / 64), dim3(64), 0, ctx->stream, D, nu + (double)N, ctx->seed,
                       ctx->sweep_host, entity_tag, draws_out);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

extern "C" int bdf_prior_pack_doubles(int D)
{
    if (D < 1 || D > BDF_MAX_D) return 0;
    const int DB = (D <= 16 ? 16 : (D <= 32 ? 32 : 64)) / 16;
    return D + DB * (DB + 1) / 2 * 4 * 64;
}

extern "C" int bdf_hyper_sample(bdf_ctx *ctx, int D, int64_t N, const double *sumU, const double *UUt,
                                const double *mu0, double b0, const double *Tinv, double nu, uint32_t entity_tag,
                                double *mu_out, double *Lambda_out, double *params_out, double *prior_pack_out,
                                const double *draws)
{
    BDF_REQUIRE(ctx && sumU && UUt && mu0 && Tinv && mu_out && Lambda_out, BDF_ERR_ARG, "bdf_hyper_sample: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_sample: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    NWArgs a;
    a.D = D; a.N = (double)N; a.sumU = sumU; a.UUt = UUt; a.mu0 = mu0; a.Tinv = Tinv; a.b0 = b0; a.nu = nu;
    a.seed = ctx->seed; a.sweep = ctx->sweep_host; a.entity_tag = entity_tag;
    a.mu_out = mu_out; a.Lambda_out = Lambda_out; a.params_out = params_out; a.pack_out = prior_pack_out; a.draws = draws;
    a.flag = ctx->flag_dev;
    if (D <= 16) hipLaunchKernelGGL(k_hyper_sample<16>, dim3(1), dim3(256), 0, ctx->stream, a);
    else if (D <= 32) hipLaunchKernelGGL(k_hyper_sample<32>, dim3(1), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_hyper_sample<64>, dim3(1), dim3(256), 0, ctx->stream, a);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}
