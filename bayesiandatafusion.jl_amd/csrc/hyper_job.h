// hyper_job.h -- device code of the hyperprior update (K6), shared by the stand-alone kernels of k_hyper.hip and by the row
// sampler's launch (k_sample_rows.hip), which can carry the previous entity's hyperprior update in a few extra workgroups.
//
//   hyper_partial<DP> : per-workgroup partial of sum_i u_i and U U' over 128 rows -- the same rank-4 MFMA update as K1's
//   hyper_reduce<DP>  : fixed-order sum of the partials (the order of k_hyper_final: 16 interleaved chains, then a butterfly)
//   nw_draw<DP>       : ConditionalNormalWishart (src/sampling.jl:116-127) + rand(::NormalWishart)
//                       (src/normal_wishart.jl:38-42) in one workgroup, random part supplied (k_hyper_draws)
#pragma once
#include "c_layout_chol.h"

namespace {

typedef double hd4 __attribute__((ext_vector_type(4)));
constexpr int HS_THREADS = 256;
constexpr int HS_ROWS = 128;         // rows per workgroup: 8 steps of 4 rows per wave at 4 waves

template <int DP>
struct HGeo {
    static constexpr int DB = DP / 16, NB = DB * (DB + 1) / 2;
    static constexpr int PSZ = NB * 4 * 64 + DB * 16;      // doubles per partial: C-layout blocks, then the column sums
    static constexpr int LD = DP + 1;
    // LDS of nw_draw (doubles): sL | sA | tri | s_muN s_mu s_rd s_sq
    // LDS of nw_draw (doubles): sL | sA | tri | s_muN s_mu s_rd s_sq | (DP <= 32) the summed U U' and column sums
    static constexpr int NW_LDS = 2 * DP * LD + Geo<DP>::TRI_D + 4 * 64 + (DP <= 32 ? DP * LD + 64 : 0);
};

struct NWArgs {
    int D;
    double N;
    const double *sumU, *UUt, *mu0, *Tinv;
    double b0, nu;
    double *mu_out, *Lambda_out, *params_out;
    const double *draws;       // Bartlett matrix (D x D row-major) + D mean normals, from k_hyper_draws
    double *pack_out;          // nullable: Lambda mu (D) then the accumulator-layout image of the reversed Lambda (K1)
    int *flag;
    uint32_t *ready;           // nullable: set to `ready_value` once the pack is written (row kernels that poll instead of waiting)
    uint32_t ready_value;
    // nullable: the sums' partials (k_hyper_partial, at most 16 of them) -- the draw then adds them itself, in k_hyper_final's
    // order, into sumU_w / UUt_w (== sumU / UUt) before it starts: one launch fewer per entity and sweep
    const double *partial;
    int nblocks;
    double *sumU_w, *UUt_w;
    int mean_ref;              // 1: mu through a second factorisation, chol(inv(Lambda) / beta_N)' z (the reference's map); 0: through the
                               // factor of Lambda the Wishart draw holds (same law, no second factorisation: nw_draw)
};

// ---- stage 1: partial sums of rows [r0, r0 + HS_ROWS) by NW waves; red: (NW-1) * PSZ doubles of LDS -----------------------
// U U' = sum over rows of u u' is the same rank-4 MFMA update as K1's: lane (j = l & 15, h = l >> 4) supplies element
// 16 I + j of row 4 s + h, straight from global memory (a coalesced 128-byte read per 16 lanes), and the lower
// block-triangle accumulates in the MFMA C layout.  The waves take every NW-th 4-row step, all of a wave's loads are
// issued before its first MFMA, and the waves' results are added in wave order.
template <int DP, int NW>
__device__ __forceinline__ void hyper_partial(int D, int64_t N, const double *__restrict__ sample, const double *__restrict__ uhat,
                                     int64_t r0, int64_t r1, double *__restrict__ p, double *red, int tid)
{
    constexpr int DB = HGeo<DP>::DB, NB = HGeo<DP>::NB, PSZ = HGeo<DP>::PSZ;
    constexpr int KS = HS_ROWS / (4 * NW);              // steps per wave and chunk of HS_ROWS rows
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, h = lane >> 4;
    hd4 acc[NB];
    double cs[DB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = hd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < DB; I++) cs[I] = 0.0;
    auto load_chunk = [&](double (&u)[KS][DB], int64_t c0) {
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const int64_t row = c0 + 4 * (wave + NW * k) + h;
#pragma unroll
            for (int I = 0; I < DB; I++) {
                const int e = 16 * I + j;
                const bool ok = row < r1 && row < N && e < D;
                const int64_t off = (ok ? row : 0) * D + (ok ? e : 0);
                const double v = sample[off] - (uhat ? uhat[off] : 0.0);
                u[k][I] = ok ? v : 0.0;
            }
        }
    };
    auto add_chunk = [&](const double (&u)[KS][DB]) {
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
    };
    if constexpr (DP <= 32) {
        // chunks in pairs, the next chunk's rows in flight under this chunk's matrix instructions (a workgroup of the chain has three to
        // five chunks and paid a load round trip for each: 3.7-5.8 us of partial sums; same sums in the same order)
        double ua[KS][DB], ub[KS][DB];
        load_chunk(ua, r0);
        for (int64_t c0 = r0; c0 < r1; c0 += 2 * HS_ROWS) {
            if (c0 + HS_ROWS < r1) load_chunk(ub, c0 + HS_ROWS);
            add_chunk(ua);
            if (c0 + HS_ROWS < r1) {
                if (c0 + 2 * HS_ROWS < r1) load_chunk(ua, c0 + 2 * HS_ROWS);
                add_chunk(ub);
            }
        }
    } else {
        for (int64_t c0 = r0; c0 < r1; c0 += HS_ROWS) {      // one chunk unless the entity is very large (workgroup count capped)
            double u[KS][DB];
            load_chunk(u, c0);
            add_chunk(u);
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
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double v = acc[b][r];
#pragma unroll
                for (int w = 0; w < NW - 1; w++) v += red[w * PSZ + (b * 4 + r) * 64 + lane];
                __hip_atomic_store(p + (b * 4 + r) * 64 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        if (lane < 16)
#pragma unroll
            for (int I = 0; I < DB; I++) {
                double v = cs[I];
#pragma unroll
                for (int w = 0; w < NW - 1; w++) v += red[w * PSZ + NB * 4 * 64 + I * 16 + lane];
                __hip_atomic_store(p + NB * 4 * 64 + I * 16 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
}

// element e of the summed partials, in the order k_hyper_final adds them: chain q takes blocks q, q+16, ..., then the
// butterfly (8, 4, 2, 1) as lane 0 of 16 sees it
__device__ __forceinline__ double hyper_sum_element(const double *partial, int psz, int nblocks, int e)
{
    double c[16];
#pragma unroll
    for (int q = 0; q < 16; q++) c[q] = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += 16) {
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (b0 + q < nblocks) c[q] += partial[(int64_t)(b0 + q) * psz + e];
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
        for (int q = 0; q < off; q++) c[q] += c[q + off];
    return c[0];
}

// scatter of a summed C-layout element to U U' (and its mirror image) / the column sums
template <int DP>
__device__ __forceinline__ void hyper_scatter(int D, int e, double s, double *sumU, double *UUt)
{
    constexpr int NB = HGeo<DP>::NB;
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
}

// ---- the Normal-Wishart draw on one workgroup (nthreads = 256 or 128); lds: HGeo<DP>::NW_LDS doubles ------------------
// In the reference's terms:
//     W    = Tinv + UU' + b0 mu0 mu0' - beta_N mu_N mu_N'         (= inv(T_N))
//     Lam  = (L_T A)(L_T A)',  L_T = chol(T_N)' lower,  A = Bartlett matrix
//     mu   = mu_N + chol(inv(Lam) / beta_N)' z
// Both "Cholesky factor of an inverse" steps come without forming the inverse (see k_hyper.hip): in index-reversed
// coordinates (~)  W~ = L~ L~',  Z~ = L~^-T (J A),  Lam~ = Z~ Z~',  Lam~ = L2~ L2~',  mu~ = mu_N~ + L2~^-T z~ / sqrt(beta_N).
// The factorisations run on wave 0 in the accumulator layout (c_layout_chol.h), the rest on all threads.
// Round 5 -- the mean WITHOUT the second factorisation (the default; a.mean_ref = 1 keeps the map above).  Lam = Z Z' with
// Z = L_T A lower triangular, so Z^-T is a square root of inv(Lam) too, and  mu = mu_N + Z^-T z / sqrt(beta_N)  has the
// reference's law N(mu_N, inv(beta_N Lam)) -- another function of the same D normals (oracle: orc_hyper_draw2, mean_map 1;
// tests/test_oracle_known_answers.py proves mean and covariance).  In the reversed coordinates  Z~^-T = L~ (J A^-T):
//     q = A^-T z   (A and z are the data-independent random part: wave 1 solves this while wave 0 factors W~)
//     mu~ = mu_N~ + L~ (J q) / sqrt(beta_N)   (one product with the factor wave 0 has just made: wave 1, beside the solve for Z~)
// which takes two dependent 32-step chains (the factorisation of Lam~ and its backward solve, ~7 us) off the iteration's
// critical path.
#ifdef BDF_HYPER_STAMPS
// diagnostic build: phase stamps of the LAST draw into a module-scope array (s_memrealtime, 100 MHz; bdf_debug_hyper_stamps) --
// 0..6 nw_draw's phases, 8 the chain's last workgroup at its start, 9 when the partial sums and the draws have arrived
__device__ unsigned long long g_hstamps[16];
#define HSTAMP(k) do { if (tid == 0) g_hstamps[(k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HSTAMP(k) do { } while (0)
#endif

template <int DP>
__device__ __forceinline__ void nw_draw(const NWArgs &a, double *lds, int tid, int nthreads)
{
    using GG = Geo<DP>;
    constexpr int LD = HGeo<DP>::LD, DB = GG::DB, NB = GG::NB;
    double *sL = lds, *sA = sL + DP * LD, *tri = sA + DP * LD, *s_muN = tri + GG::TRI_D, *s_mu = s_muN + 64,
           *s_rd = s_mu + 64, *s_sq = s_rd + 64;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, h = lane >> 4;
    const int D = a.D;
    const double beta_N = a.b0 + a.N;
    HSTAMP(0);
    bool head_done = false;
    if constexpr (DP <= 32) {
        if (a.partial && nthreads == 256) {
            // The chain's head in one pass (D <= 32, 256 threads, the sums' partials supplied): EVERY load the head needs is issued
            // before the first use -- the sixteen partials of each of the thread's elements (written through by other CUs: ~1.5 us a
            // round trip), its entries of Tinv, mu0 and the Bartlett matrix -- and the summed U U' / column sums go to LDS as well
            // as to sumU / UUt, so that mu_N and W~ are built from LDS instead of from global memory this workgroup has just written
            // (three dependent round trips: 6.3 of the chain's 27 us).  Same sums in the same order, same W.
            constexpr int PSZ = HGeo<DP>::PSZ, NE = (PSZ + 255) / 256, NW = DP * DP / 256;
            double *sUU = s_sq + 64, *s_sum = sUU + DP * LD;
            double c[NE][16];
#pragma unroll
            for (int t = 0; t < NE; t++) {
                const int e = tid + 256 * t;
#pragma unroll
                for (int q = 0; q < 16; q++) c[t][q] = (q < a.nblocks && e < PSZ) ? a.partial[(int64_t)q * PSZ + e] : 0.0;
            }
            double tinv[NW], m_lo[NW], m_hi[NW], dr[NW];
#pragma unroll
            for (int t = 0; t < NW; t++) {
                const int e = tid + 256 * t, i = e / DP, cc = e % DP;
                const int ei = D - 1 - i, ej = D - 1 - cc;
                const bool ok = ei >= 0 && ej >= 0;
                const int lo = ok ? (ei < ej ? ei : ej) : 0, hi = ok ? (ei < ej ? ej : ei) : 0;
                tinv[t] = a.Tinv[lo + (int64_t)hi * D]; m_lo[t] = a.mu0[lo]; m_hi[t] = a.mu0[hi];
                dr[t] = (ei >= 0 && cc < D) ? a.draws[ei * D + cc] : 0.0;
            }
            const double mu0_e = (tid < 64 && D - 1 - tid >= 0) ? a.mu0[D - 1 - tid] : 0.0;
#pragma unroll
            for (int t = 0; t < NE; t++) {
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
                    for (int q = 0; q < off; q++) c[t][q] += c[t][q + off];
            }
#pragma unroll
            for (int t = 0; t < NE; t++) {
                const int e = tid + 256 * t;
                if (e < PSZ) {
                    const double v = c[t][0];
                    hyper_scatter<DP>(D, e, v, a.sumU_w, a.UUt_w);
                    constexpr int NBv = HGeo<DP>::NB;
                    if (e < NBv * 4 * 64) {
                        const int b = e >> 8, r = (e >> 6) & 3, l = e & 63;
                        int I = 0;
                        while ((I + 1) * (I + 2) / 2 <= b) I++;
                        const int J = b - I * (I + 1) / 2;
                        const int row = 16 * I + (l >> 4) + 4 * r, col = 16 * J + (l & 15);
                        sUU[row * LD + col] = v;
                        if (I != J) sUU[col * LD + row] = v;
                    } else s_sum[e - NBv * 4 * 64] = v;
                }
            }
            if (D < DP)
                for (int e = tid; e < GG::TRI_D; e += nthreads) tri[e] = 0.0;
            __syncthreads();
            if (tid < 64) {
                const int e = D - 1 - tid;
                s_muN[tid] = (e >= 0) ? (a.b0 * mu0_e + s_sum[e]) / beta_N : 0.0;      // reversed: s_muN[c] = mu_N[D-1-c]
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < NW; t++) {
                const int e = tid + 256 * t, i = e / DP, cc = e % DP;
                const int ei = D - 1 - i, ej = D - 1 - cc;
                double w = (i == cc) ? 1.0 : 0.0;
                if (ei >= 0 && ej >= 0) {
                    const int lo = ei < ej ? ei : ej, hi = ei < ej ? ej : ei;
                    w = tinv[t] + sUU[lo * LD + hi] + a.b0 * m_lo[t] * m_hi[t] - beta_N * s_muN[D - 1 - lo] * s_muN[D - 1 - hi];
                    if (a.params_out) a.params_out[D + ei + (int64_t)ej * D] = w;
                }
                sL[i * LD + cc] = w;
                sA[i * LD + cc] = dr[t];                                              // sA[i][c] = A[D-1-i][c]
            }
            if (a.params_out && tid < DP && D - 1 - tid >= 0) a.params_out[D - 1 - tid] = s_muN[tid];
            __syncthreads();
            head_done = true;
        }
    }
    if (!head_done) {
    if (a.partial) {
        // stage 2 of the sums (k_hyper_final's order: chain q = partial q, then the butterfly 8, 4, 2, 1); all loads of an
        // element in flight together
        constexpr int PSZ = HGeo<DP>::PSZ;
        for (int e = tid; e < PSZ; e += nthreads) {
            double c[16];
#pragma unroll
            for (int q = 0; q < 16; q++) c[q] = q < a.nblocks ? a.partial[(int64_t)q * PSZ + e] : 0.0;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
                for (int q = 0; q < off; q++) c[q] += c[q + off];
            hyper_scatter<DP>(D, e, c[0], a.sumU_w, a.UUt_w);
        }
        __syncthreads();                            // (workgroup-scope release / acquire: the sums are read below)
    }
    if (D < DP)                                     // the padding's entries of the packed factor are never stored: zero (backward_batch)
        for (int e = tid; e < GG::TRI_D; e += nthreads) tri[e] = 0.0;
    if (tid < 64) {
        const int e = D - 1 - tid;
        s_muN[tid] = (e >= 0) ? (a.b0 * a.mu0[e] + a.sumU[e]) / beta_N : 0.0;      // reversed: s_muN[c] = mu_N[D-1-c]
    }
    __syncthreads();
    // ---- W~ (Symmetric(): upper triangle) into sL, the reversed-row Bartlett matrix A~ = J A into sA: all threads
    for (int e = tid; e < DP * DP; e += nthreads) {
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
        sA[i * LD + c] = (ei >= 0 && c < D) ? a.draws[ei * D + c] : 0.0;          // sA[i][c] = A[D-1-i][c]
    }
    if (a.params_out && tid < DP && D - 1 - tid >= 0) a.params_out[D - 1 - tid] = s_muN[tid];
    __syncthreads();
    }
    HSTAMP(1);

    // the factorisation of the matrix in sL on wave 0; pivots' reciprocals and square roots to s_rd / s_sq
    auto factor_sL = [&](double &dv_out) {
        double A[NB * 4], bv[DB], ts[DB];
#pragma unroll
        for (int I = 0; I < DB; I++)
#pragma unroll
            for (int J = 0; J <= I; J++)
#pragma unroll
                for (int r = 0; r < 4; r++) A[GG::blk(I, J) * 4 + r] = sL[(16 * I + h + 4 * r) * LD + 16 * J + j];
#pragma unroll
        for (int J = 0; J < DB; J++) { bv[J] = 0.0; ts[J] = 0.0; }
        wave_sync();
        factor_all<DP>(A, bv, ts, tri, j, h, D, std::make_integer_sequence<int, DP - 1>{});
        wave_sync();
        const typename GG::ColRT cr = GG::col_rt(lane < DP ? lane : 0);
        double dv = 1.0;
        if (lane < D) dv = tri[cr.cbase + (lane & 3) * cr.nr4];
        if (!(dv > 0.0)) atomicOr_system(a.flag, 2);
        dv_out = dv;
        return cr;
    };

    if (wave == 0) {
        double dv;
        factor_sL(dv);
        s_rd[lane] = fast_rcp(dv);
        s_sq[lane] = dv * fast_rsqrt(dv);
    } else if (wave == 1 && !a.mean_ref) {
        // q = A^-T z by backward substitution over the Bartlett matrix (lower triangular; sA[i][c] = A[D-1-i][c], zero rows on the
        // padding): lane c holds z_c - sum_(k > c) A_kc q_k; DP unconditional steps, natural row k = D - 1 - i (negative: padding, no-op)
        double zi = (lane < D) ? a.draws[D * D + lane] : 0.0;
        const double rai = (lane < D) ? fast_rcp(sA[(D - 1 - lane) * LD + lane]) : 1.0;
        if constexpr (DP <= 32) {
#pragma unroll
            for (int i = 0; i < DP; i++) {
                const int k = D - 1 - i;
                const double qk = __shfl(zi * rai, k & 63);        // (lane k's entry is final when its turn comes)
                const double aki = sA[i * LD + lane];
                zi = (lane < k) ? fma(-aki, qk, zi) : zi;
            }
        } else {
            // (D = 64: not unrolled -- the kernel's registers are the solve for Z~'s, 64 doubles per lane)
#pragma unroll 1
            for (int i = 0; i < DP; i++) {
                const int k = D - 1 - i;
                const double qk = __shfl(zi * rai, k & 63);
                const double aki = sA[i * LD + lane];
                zi = (lane < k) ? fma(-aki, qk, zi) : zi;
            }
        }
        s_mu[lane] = (lane < D) ? zi * rai : 0.0;                   // q, natural order (s_mu is free until the mean is written)
    }
    __syncthreads();
    HSTAMP(2);

    // ---- Z~ = L~^-T A~  <=>  Lt' Z~ = diag(sqrt(d)) A~ : one thread per column, backward substitution over the packed factor
    if (tid < DP) {
        double z[DP];
#pragma unroll
        for (int i = DP - 1; i >= 0; i--) {
            if (i >= D) { z[i] = 0.0; continue; }            // padding: identity
            constexpr int dummy = 0; (void)dummy;
            // four interleaved partial sums (fixed assignment m % 4): four short dependency chains instead of one long one
            double s4[4] = {s_sq[i] * sA[i * LD + tid], 0.0, 0.0, 0.0};
            const int cb = GG::col_base(i), nr4 = GG::col_rows(i) / 4, r16 = GG::col_first(i);
#pragma unroll
            for (int m = i + 1; m < DP; m++) s4[m & 3] = fma(-tri[cb + (m & 3) * nr4 + (m - r16) / 4], z[m], s4[m & 3]);
            z[i] = ((s4[0] + s4[1]) + (s4[2] + s4[3])) * s_rd[i];
        }
#pragma unroll
        for (int i = 0; i < DP; i++) sA[i * LD + tid] = z[i];
    } else if (wave == 1 && !a.mean_ref) {
        // mu~ = mu_N~ + L~ v / sqrt(beta_N), v = J q; L~[i][k] = Lt[i][k] / sqrt(d_k) (k < i), L~[i][i] = sqrt(d_i): lane = row i
        // (the padding's entries of the packed factor are zero, its pivots 1)
        const int i = lane;
        double acc4[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (DP <= 32) {
#pragma unroll
            for (int k = 0; k < DP - 1; k++) {
                const int cb = GG::col_base(k), nr4 = GG::col_rows(k) / 4, r16 = GG::col_first(k);
                const double l = (i > k && i < DP) ? tri[cb + (i & 3) * nr4 + (i - r16) / 4] : 0.0;
                const double vk = (k < D) ? s_mu[(D - 1 - k) & 63] : 0.0;
                acc4[k & 3] = fma(l * (s_rd[k] * s_sq[k]), vk, acc4[k & 3]);
            }
        } else {
            double accs = 0.0;
#pragma unroll 1
            for (int k = 0; k < D - 1; k++) {
                const typename GG::ColRT cr = GG::col_rt(k);
                const double l = (i > k && i < DP) ? tri[cr.cbase + (i & 3) * cr.nr4 + (i >> 2) - cr.q] : 0.0;
                accs = fma(l * (s_rd[k] * s_sq[k]), s_mu[D - 1 - k], accs);
            }
            acc4[0] = accs;
        }
        const double lv = (i < D) ? ((acc4[0] + acc4[1]) + (acc4[2] + acc4[3])) + s_sq[i] * s_mu[(D - 1 - i) & 63] : 0.0;
        const double mu_c = s_muN[lane] + lv / sqrt(beta_N);
        wave_sync();                                               // every lane has read q before the mean replaces it
        if (i < D) a.mu_out[D - 1 - i] = mu_c;
        s_mu[lane] = (i < D) ? mu_c : 0.0;                          // reversed: s_mu[c] = mu[D-1-c]
    }
    __syncthreads();
    HSTAMP(3);

    // ---- Lam~ = Z~ Z~' on the matrix cores: block (I, J >= ... all blocks of the lower block-triangle, mirrored; the diagonal
    // blocks multiply the same pairs in the same order for (i,j) and (j,i), so the result is exactly symmetric).  Stored
    // reversed in sL (identity on the padding) and natural in Lambda_out.
    {
        const int nw = nthreads / 64;
        for (int b = wave; b < NB; b += nw) {
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= b) I++;
            const int J = b - I * (I + 1) / 2;
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < DP / 4; s++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[(16 * I + j) * LD + 4 * s + h], sA[(16 * J + j) * LD + 4 * s + h], acc,
                                                           0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 16 * I + h + 4 * r, c2 = 16 * J + j;
                const int ei = D - 1 - i, ej = D - 1 - c2;
                double v = acc[r];
                if (ei < 0 || ej < 0) v = (i == c2) ? 1.0 : 0.0;
                else {
                    a.Lambda_out[ei + (int64_t)ej * D] = v;
                    if (I != J) a.Lambda_out[ej + (int64_t)ei * D] = v;
                }
                sL[i * LD + c2] = v;
                if (I != J) sL[c2 * LD + i] = v;
            }
        }
    }
    __syncthreads();
    HSTAMP(4);

    // ---- (the reference's map only) mu~ = mu_N~ + L2~^-T z~ / sqrt(beta_N), Lam~ = L2~ L2~'
    if (wave == 0 && a.mean_ref) {
        double dv;
        const typename GG::ColRT cr = factor_sL(dv);
        const int ej = D - 1 - lane;
        const double rdv = fast_rcp(dv);
        double yh = (lane < D) ? a.draws[D * D + ej] * (dv * fast_rsqrt(dv)) : 0.0;
        unsigned colq[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            colq[q] = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(tri + cr.cbase + q * cr.nr4 - cr.q);
        backward_all<DP>(yh, rdv, colq, std::make_integer_sequence<int, DP / 16>{});
        const double mu_c = s_muN[lane] + (yh * rdv) / sqrt(beta_N);
        if (lane < D) a.mu_out[ej] = mu_c;
        s_mu[lane] = (lane < D) ? mu_c : 0.0;                      // reversed: s_mu[c] = mu[D-1-c]
    }
    HSTAMP(5);
    if (a.pack_out == nullptr) return;
    __syncthreads();
    // ---- what the row sampler needs of (mu, Lambda), written here so that it needs no pre-launch of its own:
    // Lambda mu (same products in the same order as k_prior of k_sample_rows.hip: bit-identical), and Lam~ in
    // the MFMA accumulator layout [block * 4 + r][lane] (identity on the padding -- exactly what sL holds)
    for (int e = tid >> 3; e < D; e += nthreads / 8) {            // eight lanes per entry: lane part p adds i = p, p+8, ...
        const int part = tid & 7;
        double v = 0.0;
        for (int i = part; i < D; i += 8) v = fma(sL[(D - 1 - e) * LD + (D - 1 - i)], s_mu[D - 1 - i], v);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (part == 0) __hip_atomic_store(a.pack_out + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int e = wave; e < NB * 4; e += nthreads / 64) {
        const int b = e >> 2, r = e & 3;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= b) I++;
        const int J = b - I * (I + 1) / 2;
        __hip_atomic_store(a.pack_out + D + e * 64 + lane, sL[(16 * I + (lane >> 4) + 4 * r) * LD + 16 * J + (lane & 15)], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);      // write-through: a polling row kernel reads it past its L2
    }
    HSTAMP(6);
}

}  // namespace
