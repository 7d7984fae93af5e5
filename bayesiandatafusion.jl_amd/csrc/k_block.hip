// k_block.hip -- sample_users_blocked (src/sampling.jl:236-249): the users of a Block all observed the same items, so they
// share ONE conditional covariance:
//     covar = inv(Lambda_u + alpha MM MM'),  mu = covar (alpha MM Yma + Lambda_u mu_u),  sample = chol(covar)' z + mu
// with MM = sample_mt[:, block.vx].  One wave accumulates P = Lambda_u + alpha MM MM' on the matrix cores and factors it once
// (k_block_factor: the row sampler's accumulator-layout factorisation, index-reversed coordinates as there, so that the
// result is the reference's function of z: k_sample_rows.hip); then one wave per user forms its right-hand side (a gather of
// the nv item rows weighted by the user's column of Yma), runs the forward and backward substitution against the shared
// packed factor (one LDS copy per workgroup) and adds its own normals (k_block_users).  The row sampler would factor the same
// matrix once per user.
#include "bdf_common.h"
#include "c_layout_chol.h"

namespace {

// P~ = image of the reversed Lambda (prior_c, from k_prior) + alpha sum_o w_o w_o', factored; the packed factor (TRI_D doubles)
// and the pivots' reciprocals / square roots (3 x DP doubles: d, 1/d, sqrt(d)) go to global memory
template <int DP>
__global__ __launch_bounds__(64) void k_block_factor(int D, int64_t nv, const int32_t *__restrict__ vx, const double *__restrict__ factor,
                                                     double alpha, const double *__restrict__ prior_c, double *__restrict__ fac_out,
                                                     double *__restrict__ piv_out, int *flag)
{
    using GG = Geo<DP>;
    constexpr int DB = GG::DB, NB = GG::NB;
    __shared__ __attribute__((aligned(16))) double tri[GG::TRI_D];
    const int lane = threadIdx.x, j = lane & 15, h = lane >> 4;
    d4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    for (int64_t o0 = 0; o0 < nv; o0 += 8) {              // two k-steps of 4 items per trip
        double w[2][DB];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int64_t o = o0 + 4 * k + h;
            const bool ok = o < nv;
            const double *f = factor + (int64_t)vx[ok ? o : 0] * D;
#pragma unroll
            for (int I = 0; I < DB; I++) {
                const int ec = D - 1 - (16 * I + j);
                w[k][I] = (ok && ec >= 0) ? f[ec] : 0.0;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int b = 0;
#pragma unroll
            for (int I = 0; I < DB; I++)
#pragma unroll
                for (int J = 0; J <= I; J++) {
                    acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[k][I], w[k][J], acc[b], 0, 0, 0);
                    b++;
                }
        }
    }
    double A[NB * 4], bv[DB], ts[DB];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) A[b * 4 + r] = fma(acc[b][r], alpha, prior_c[(b * 4 + r) * 64 + lane]);
#pragma unroll
    for (int J = 0; J < DB; J++) { bv[J] = 0.0; ts[J] = 0.0; }
    if (D < DP) zero_packed_factor<DP>(tri, lane);
    factor_all<DP>(A, bv, ts, tri, j, h, D, std::make_integer_sequence<int, DP - 1>{});
    wave_sync();
    for (int e = lane; e < GG::TRI_D; e += 64) fac_out[e] = tri[e];
    if (lane < DP) {
        const typename GG::ColRT cr = GG::col_rt(lane);
        double dv = 1.0;
        if (lane < D) dv = tri[cr.cbase + (lane & 3) * cr.nr4];
        if (!(dv > 0.0)) { atomicOr_system(flag, 1); dv = 1.0; }
        piv_out[lane] = dv;
        piv_out[DP + lane] = fast_rcp(dv);
        piv_out[2 * DP + lane] = dv * fast_rsqrt(dv);
    }
}

// one wave per user: b~ = (Lambda mu)~ + alpha sum_o Yma[o][u] w_o~ ; t = forward substitution; yh = t + z sqrt(d); backward; store
template <int DP>
__global__ __launch_bounds__(256) void k_block_users(int D, int64_t nu, int64_t nv, const int32_t *__restrict__ vx,
                                                     const double *__restrict__ factor, const double *__restrict__ Yma, double alpha,
                                                     const double *__restrict__ prior_b, const double *__restrict__ fac, const double *__restrict__ piv,
                                                     uint64_t seed, uint32_t sweep, uint32_t entity_tag, double *__restrict__ out)
{
    using GG = Geo<DP>;
    __shared__ __attribute__((aligned(16))) double tri[GG::TRI_D];
    __shared__ double s_rd[DP];
    for (int e = threadIdx.x; e < GG::TRI_D; e += 256) tri[e] = fac[e];
    if (threadIdx.x < DP) s_rd[threadIdx.x] = piv[DP + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= nu) return;
    const int c = lane & (DP - 1);                        // reversed coordinate handled by this lane (lanes >= DP mirror: second item of a pair)
    const int half = (DP < 64) ? (lane / DP) : 0;         // DP = 16: four items per trip, 32: two, 64: one
    constexpr int PER = 64 / DP;
    const int ec = D - 1 - c;
    double z = 0.0;
    if (lane < D) z = bdf_normal(seed, sweep, BDF_P_ROW, entity_tag, (uint64_t)u, D - 1 - lane);
    double b = 0.0;
    for (int64_t o0 = 0; o0 < nv; o0 += PER) {
        const int64_t o = o0 + half;
        if (o < nv && ec >= 0) b = fma(factor[(int64_t)vx[o] * D + ec], Yma[o + u * nv], b);
    }
    if (PER >= 4) b += __shfl_xor(b, 16);
    if (PER >= 4) b += __shfl_xor(b, 32);
    if (PER == 2) b += __shfl_xor(b, 32);
    b = (lane < D) ? fma(b, alpha, prior_b[ec]) : 0.0;
    // forward substitution against the unscaled packed factor: t_i final at step i, rows c > i updated with -Lt[c][i] t_i / d_i
    for (int i = 0; i < D - 1; i++) {
        const typename GG::ColRT cr = GG::col_rt(i);
        const double ti = readlane_f64(b, i) * s_rd[i];
        if (lane > i && lane < D) b = fma(-tri[cr.cbase + (lane & 3) * cr.nr4 + (lane >> 2) - cr.q], ti, b);
    }
    const typename GG::ColRT cr = GG::col_rt(lane < DP ? lane : 0);
    const double dv = (lane < DP) ? piv[lane] : 1.0, rdv = (lane < DP) ? piv[DP + lane] : 1.0, sq = (lane < DP) ? piv[2 * DP + lane] : 1.0;
    double yh = (lane < D) ? fma(z, sq, b) : 0.0;
    (void)dv;
    unsigned colq[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        colq[q] = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(tri + cr.cbase + q * cr.nr4 - cr.q);
    backward_all<DP>(yh, rdv, colq, std::make_integer_sequence<int, DP / 16>{});
    if (lane < D) out[u * D + (D - 1 - lane)] = yh * rdv;
}

}  // namespace

// k_prior of k_sample_rows.hip: Lambda mu and the accumulator-layout image of the reversed Lambda
int bdf_prior_image(bdf_ctx *ctx, int D, const double *Lambda, const double *mu, double *out_b, double *out_c);

extern "C" int bdf_sample_block(bdf_ctx *ctx, int D, int64_t nu, int64_t nv, const int32_t *vx_dev, const double *Yma,
                                const double *factor, double alpha, const double *mu, const double *Lambda, uint32_t entity_tag,
                                double *out)
{
    BDF_REQUIRE(ctx && mu && Lambda && (nu == 0 || out) && (nv == 0 || (vx_dev && factor)) && (nu * nv == 0 || Yma), BDF_ERR_ARG,
                "bdf_sample_block: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D && nu >= 0 && nv >= 0, BDF_ERR_ARG, "bdf_sample_block: bad size");
    if (nu == 0) return BDF_OK;
    const int DP = D <= 16 ? 16 : (D <= 32 ? 32 : 64);
    const int DB = DP / 16, nimg = DB * (DB + 1) / 2 * 4;
    const size_t tri_d = DP == 16 ? Geo<16>::TRI_D : (DP == 32 ? Geo<32>::TRI_D : Geo<64>::TRI_D);
    void *sc;
    int rc = bdf_scratch(ctx, ((size_t)D + (size_t)nimg * 64 + tri_d + 3 * (size_t)DP) * sizeof(double), &sc);
    if (rc) return rc;
    double *pb = (double *)sc, *pc = pb + D, *fac = pc + (size_t)nimg * 64, *piv = fac + tri_d;
    if ((rc = bdf_prior_image(ctx, D, Lambda, mu, pb, pc))) return rc;
    const dim3 ug((unsigned)((nu + 3) / 4));
#define BLOCK(DPV)                                                                                                      \
    do {                                                                                                                \
        hipLaunchKernelGGL(k_block_factor<DPV>, dim3(1), dim3(64), 0, ctx->stream, D, nv, vx_dev, factor, alpha, (const double *)pc, fac, piv, ctx->flag_dev); \
        hipLaunchKernelGGL(k_block_users<DPV>, ug, dim3(256), 0, ctx->stream, D, nu, nv, vx_dev, factor, Yma, alpha, (const double *)pb, \
                           (const double *)fac, (const double *)piv, ctx->seed, ctx->sweep_host, entity_tag, out);     \
    } while (0)
    if (DP == 16) BLOCK(16); else if (DP == 32) BLOCK(32); else BLOCK(64);
#undef BLOCK
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}
