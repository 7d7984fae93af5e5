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
// LDS needs per wave (doubles): fb[64] broadcast row, piv[64] pivots then 1/pivots, img[DP * (DP + 1)] transposition.
#pragma once
#include "bdf_common.h"

__device__ inline void wave_sync()
{
    // orders this wave's LDS writes before its later LDS reads (the LDS pipe is in-order per wave; this only stops the
    // compiler from moving accesses across it)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ inline double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

__device__ inline double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(y * 0.5, e, y);
}

// In : col[i] = A[i][c] (symmetric, all rows).
// Out: col[k] = Ah[c][k] for k < c, 0 for k >= c (strict lower part of row c, masked); p_own = pivot p_c;
//      piv[lane] = 1 / p_c (for the solves).  Returns true (in every lane of the group) if a pivot was not positive.
template <int DP>
__device__ inline bool wl_factor(double (&col)[DP], double &p_own, double *fb, double *piv, int lane)
{
    const int grp = lane / DP, c = lane % DP;
    double *row = fb + grp * DP;
    bool notpd = false;
    p_own = 1.0;
#pragma unroll
    for (int k = 0; k < DP; k++) {
        row[c] = col[k];                                  // A[k][c] = A[c][k]
        wave_sync();
        const double pk = row[k];
        if (!(pk > 0.0)) notpd = true;
        const double g = col[k] * fast_rcp(pk);           // A[c][k] / A[k][k]
#pragma unroll
        for (int i = k + 1; i < DP; i++) col[i] = fma(-row[i], g, col[i]);
        p_own = (c == k) ? pk : p_own;
        wave_sync();                                      // row[] is rewritten by the next step (compiler barrier only)
    }
    piv[lane] = fast_rcp(p_own);
#pragma unroll
    for (int k = 0; k < DP; k++) col[k] = (k < c) ? col[k] : 0.0;
    wave_sync();
    return notpd;
}

__device__ inline double wl_bcast(double v, int src_lane)
{
    return __shfl(v, src_lane);
}

// forward solve Ah wh = b on masked rows (lane c holds row c).  Returns b'_c = wh_c * p_c (the reduced right-hand side).
template <int DP>
__device__ inline double wl_forward(const double (&rowm)[DP], double b, const double *piv, int lane)
{
    const int base = (lane / DP) * DP;
#pragma unroll
    for (int k = 0; k < DP; k++) {
        const double wk = wl_bcast(b, base + k) * piv[base + k];
        b = fma(-rowm[k], wk, b);
    }
    return b;
}

// masked rows -> masked columns through the image (img: DP * (DP+1) doubles), one group at a time.
// in: rowm[k] = Ah[c][k] (k < c, else 0).  out: colm[i] = Ah[i][c] (i > c, else 0).
template <int DP>
__device__ inline void wl_transpose(double (&col)[DP], double *img, int lane)
{
    constexpr int G = 64 / DP, LD = DP + 1;
    const int grp = lane / DP, c = lane % DP;
#pragma unroll
    for (int g = 0; g < G; g++) {
        wave_sync();
        if (grp == g) {
#pragma unroll
            for (int k = 0; k < DP; k++) img[c * LD + k] = col[k];
        }
        wave_sync();
        if (grp == g) {
#pragma unroll
            for (int i = 0; i < DP; i++) col[i] = img[i * LD + c];
        }
    }
    wave_sync();
}

// backward solve Ah' x = yh on masked columns (lane c holds column c).  Returns x_c.
template <int DP>
__device__ inline double wl_backward(const double (&colm)[DP], double yh, const double *piv, int lane)
{
    const int base = (lane / DP) * DP;
#pragma unroll
    for (int i = DP - 1; i >= 0; i--) {
        const double xi = wl_bcast(yh, base + i) * piv[base + i];
        yh = fma(-colm[i], xi, yh);
    }
    return yh * piv[lane];
}
