// wave_linalg.h -- D x D (D <= 64) symmetric positive-definite linear algebra on ONE wavefront.
//
// Layout: lane j owns column j of the matrix in registers col[0..DP-1] (col[i] = A[i][j]); the matrix is kept
// fully symmetric so that row k of a lane's column doubles as A[j][k].  Every loop is unrolled so that register
// indices are static; cross-lane traffic is v_readlane broadcasts of a wave-uniform lane (no LDS).
#pragma once
#include "bdf_common.h"

constexpr int WL_TLD = 33;   // leading dimension of the LDS transpose buffer (32 columns per pass + 1 pad)

// Cholesky A = L L'.  On exit lane j holds ROW j of L: col[k] = L[j][k] for k <= j (col[k], k > j: undefined),
// rinv_own = 1 / L[j][j].  Returns true in every lane if a pivot was not positive.
template <int DP>
__device__ inline bool wl_chol_rows(double (&col)[DP], double &rinv_own, int lane)
{
    bool notpd = false;
    rinv_own = 1.0;
#pragma unroll
    for (int k = 0; k < DP; k++) {
        const double pk = readlane_f64(col[k], k);
        if (!(pk > 0.0)) notpd = true;
        const double rinv = 1.0 / sqrt(pk);
        const double f = col[k] * rinv;            // lane j: A[k][j]/sqrt(pk) = L[j][k]  (j >= k)
        col[k] = f;
        if (lane == k) rinv_own = rinv;
#pragma unroll
        for (int i = k + 1; i < DP; i++) col[i] = fma(-readlane_f64(f, i), f, col[i]);
    }
    return notpd;
}

// forward substitution L w = b with lane j holding row j of L (wl_chol_rows layout) and b_j; returns w_j
template <int DP>
__device__ inline double wl_fwd_rows(const double (&col)[DP], double rinv_own, double bj, int lane)
{
#pragma unroll
    for (int k = 0; k < DP; k++) {
        const double wk = readlane_f64(bj, k) * readlane_f64(rinv_own, k);
        if (lane > k) bj = fma(-col[k], wk, bj);
        else if (lane == k) bj = wk;
    }
    return bj;
}

// rows -> columns through LDS (tb: DP * WL_TLD doubles), 32 columns per pass.
// in : lane j holds row j (col[k] = L[j][k], k <= j).  out: lane j holds column j (col[i] = L[i][j], i >= j).
// Must be called by all 64 lanes of a single-wave workgroup.
template <int DP>
__device__ inline void wl_rows_to_cols(double (&col)[DP], double *tb, int lane)
{
#pragma unroll
    for (int pass = 0; pass < (DP + 31) / 32; pass++) {
        const int k0 = pass * 32;
        __syncthreads();
        if (lane < DP) {
#pragma unroll
            for (int k = 0; k < 32 && k0 + k < DP; k++) tb[lane * WL_TLD + k] = col[k0 + k];
        }
        __syncthreads();
        if (lane >= k0 && lane < k0 + 32 && lane < DP) {
#pragma unroll
            for (int i = 0; i < DP; i++) col[i] = tb[i * WL_TLD + (lane - k0)];
        }
    }
}

// backward substitution L' x = y with lane j holding COLUMN j of L (col[i] = L[i][j], i >= j) and y_j; returns x_j
template <int DP>
__device__ inline double wl_bwd_cols(const double (&col)[DP], double rinv_own, double yj, int lane)
{
#pragma unroll
    for (int i = DP - 1; i >= 0; i--) {
        const double xi = readlane_f64(yj, i) * readlane_f64(rinv_own, i);
        if (lane < i) yj = fma(-col[i], xi, yj);
        else if (lane == i) yj = xi;
    }
    return yj;
}
