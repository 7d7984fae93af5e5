// wave_linalg.h -- D x D (D <= 64) symmetric positive-definite linear algebra on ONE wavefront, G = 64/DP matrices at a
// time (DP = 16, 32, 64 is the padded dimension): lane = grp * DP + c, lane (grp, c) holds column c of matrix grp in
// registers, col[i] = A[i][c].  The matrix is kept fully symmetric so that row k of a lane's column doubles as A[c][k].
//
// Factorisation A = Ah diag(1/p) Ah' with Ah = L diag(sqrt(p)) (L the Cholesky factor, p the pivots): kept UNSCALED so
// that no square root sits on the step-to-step critical path.  Step k broadcasts row k of the current Schur complement
// through LDS (one ds_write_b64 + broadcast reads) and updates every trailing row with one fma per element.
// The triangular solves are written in terms of Ah and 1/p:
//     L w = b   <=>  Ah wh = b,   wh = w / sqrt(p)
//     L' x = y  <=>  Ah' x = yh,  yh = y * sqrt(p)
// and run on MASKED columns (strict lower part, zeros elsewhere) so that their inner step is one unconditional fma.
//
// LDS needs per wave: WL<DP>::tri_doubles(share) doubles for the packed factor.
#pragma once
#include "bdf_common.h"

#ifndef WL_CH
#define WL_CH 8
#endif

__device__ __forceinline__ void wave_sync()
{
    // orders this wave's LDS writes before its later LDS reads (the LDS pipe is in-order per wave; this only stops the
    // compiler from moving accesses across it)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1 / x from v_rcp_f64 (relative error 3.9e-8 as measured) and Newton steps: two give 1e-16; one gives 1.5e-15, which is what
// the factorisations use for their pivots (BDF_RCP_STEPS) -- the FP64 pipe is what bounds the row kernel, and the second step
// is 62 of its ~870 fp64 instructions per row
#ifndef BDF_RCP_STEPS
#define BDF_RCP_STEPS 1
#endif
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    if (BDF_RCP_STEPS > 1) {
        e = fma(-x, y, 1.0);
        y = fma(y, e, y);
    }
    return y;
}

__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(y * 0.5, e, y);
}

// In : col[i] = A[i][c] (symmetric, all rows).
// Out: col[k] = Ah[c][k] for k < c, 0 for k >= c (strict lower part of row c, masked); p_own = pivot p_c,
//      rp_own = 1 / p_c; tri = the factor by columns in packed form: column k (entries Ah[i][k], i >= k, Ah[k][k] = p_k)
//      at tri[grp * TRI + off(k) + (i - k)], off(k) = k DP - k (k-1) / 2, TRI = DP (DP+1) / 2.  Step k's broadcast row IS
//      column k of the factor, so storing it costs nothing and the backward solve needs no transposition.
//      tri needs G * TRI + 64 doubles (SHARE: every group holds the same matrix and they share one copy: TRI + 64).
// Returns true (in every lane of the group) if a pivot was not positive.
template <int DP>
struct WL {
    static constexpr int G = 64 / DP;
    static constexpr int TRI = DP * (DP + 1) / 2;
    __host__ __device__ static constexpr int off(int k) { return k * DP - k * (k - 1) / 2; }
    static constexpr int tri_doubles(bool share) { return (share ? 1 : G) * TRI + 64; }
};

template <int DP, bool SHARE = false>
__device__ __forceinline__ bool wl_factor(double (&col)[DP], double &p_own, double &rp_own, double *tri, int lane)
{
    using W = WL<DP>;
    const int grp = lane / DP, c = lane % DP;
    double *tg = tri + (SHARE ? 0 : grp * W::TRI);
    double *dummy = tri + (SHARE ? 1 : W::G) * W::TRI + lane;
    bool notpd = false;
    p_own = 1.0;
#pragma unroll
    for (int k = 0; k < DP; k++) {
        double *rk = tg + W::off(k) - k;                  // rk[i] = entry i of column k (i >= k)
        double *dst = (c >= k) ? rk + c : dummy;          // lanes c < k are done: they park their store
        *dst = col[k];                                    // A[k][c] = A[c][k]
        wave_sync();
        const double pk = rk[k];
        if (!(pk > 0.0)) notpd = true;
        const double g = col[k] * fast_rcp(pk);           // A[c][k] / A[k][k]
        // trailing rows in chunks of WL_CH: a compiler-only memory barrier after each chunk's reads keeps at most two
        // chunks of broadcast values live (otherwise every read of the step is hoisted: +2 (DP-k) registers)
        {
            constexpr int CH = WL_CH;
            double cur[CH], nxt[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) cur[u] = (k + 1 + u < DP) ? rk[k + 1 + u] : 0.0;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i0 = k + 1; i0 < DP; i0 += CH) {
#pragma unroll
                for (int u = 0; u < CH; u++) nxt[u] = (i0 + CH + u < DP) ? rk[i0 + CH + u] : 0.0;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < CH; u++)
                    if (i0 + u < DP) col[i0 + u] = fma(-cur[u], g, col[i0 + u]);
#pragma unroll
                for (int u = 0; u < CH; u++) cur[u] = nxt[u];
            }
        }
        p_own = (c == k) ? pk : p_own;
    }
    rp_own = fast_rcp(p_own);
#pragma unroll
    for (int k = 0; k < DP; k++) col[k] = (k < c) ? col[k] : 0.0;
    wave_sync();
    return notpd;
}

__device__ __forceinline__ double wl_bcast(double v, int src_lane)
{
    return __shfl(v, src_lane);
}

// forward solve Ah wh = b on masked rows (lane c holds row c; rp_own = 1 / p_c).
// Returns b'_c = wh_c * p_c (the reduced right-hand side).
template <int DP>
__device__ __forceinline__ double wl_forward(const double (&rowm)[DP], double b, double rp_own, int lane)
{
    const int base = (lane / DP) * DP;
#pragma unroll
    for (int k = 0; k < DP; k++) {
        const double wk = wl_bcast(b * rp_own, base + k);     // lane k's b is final at step k
        b = fma(-rowm[k], wk, b);
    }
    return b;
}

// backward solve Ah' x = yh reading the factor's columns from tri (lane c uses column c).  Returns x_c.
template <int DP, bool SHARE = false>
__device__ __forceinline__ double wl_backward(const double *tri, double yh, double rp_own, int lane)
{
    using W = WL<DP>;
    const int grp = lane / DP, c = lane % DP, base = grp * DP;
    const double *colc = tri + (SHARE ? 0 : grp * W::TRI) + (c * DP - c * (c - 1) / 2) - c;   // colc[i] = Ah[i][c], i >= c
    const double *zero = tri + (SHARE ? 1 : W::G) * W::TRI + lane;
    *(double *)zero = 0.0;          // wl_factor parked arbitrary values in these slots
    wave_sync();
#pragma unroll
    for (int i = DP - 1; i >= 0; i--) {
        const double xi = wl_bcast(yh * rp_own, base + i);    // lane i's yh is final at step i
        const double *src = (i > c) ? colc + i : zero;
        yh = fma(-(*src), xi, yh);
    }
    return yh * rp_own;
}
