// dpp_rows32.h -- a 32 x 32 (or smaller) symmetric positive-definite system per ROW OF 16 LANES, four systems per wavefront:
// lane j of a lane row holds COLUMNS j and 16 + j of the system in full (A0[i] = entry (i, j), A1[i] = entry (i, 16 + j); entry 32 of
// each: the right-hand side) -- the unfinished part stays symmetric, so the multiplier of a column in step k is the lane's own
// entry of row k: no transposition, no LDS.  The LDL' factorisation with the forward solve riding along as row 32 and the
// backward solve are v_fmac_f64_dpp row_newbcast instructions (lane k % 16 of each lane row is the source of step k).  Used by
// k_rows_col (k_rows_col.hip: K1c, four rows per wave from the first observation on).
#pragma once
#include "dpp_rows16.h"

namespace {

// offset (doubles) in a partial slot / in the prior's image of element (i, 16 s + j) of the reversed system, DP = 32:
// blocks (0,0), (1,0), (1,1) in the accumulator layout -- lane (jj, hh), register r of block (I, J) is element
// (16 I + hh + 4 r, 16 J + jj) at [(blk * 4 + r) * 64 + 16 hh + jj]; the block above the diagonal is read from its mirror image
template <int S, int I>
__device__ __forceinline__ int sys_off(int j)
{
    constexpr int IB = I / 16, ii = I % 16;
    if constexpr (IB >= S) {
        constexpr int blk = IB * (IB + 1) / 2 + S;
        return (blk * 4 + (ii >> 2)) * 64 + 16 * (ii & 3) + j;
    } else {
        return (1 * 4 + (j >> 2)) * 64 + 16 * (j & 3) + ii;       // element (16 + j, i) of block (1, 0)
    }
}

// ---- a rank-1 update v v' of the three stored blocks (0,0), (1,0), (1,1): lane j holds v0 = v_j and v1 = v_(16+j) (k_rows_col: an observation;
// k_rows_lr32: an element of the two observations a lane streams) ----
template <int DR, int I>
__device__ __forceinline__ void col_rank1(double (&A0)[33], double (&A1)[33], double v0, double v1)
{
    if constexpr (I < 16) {
        if constexpr (I == 0) fm1<0>(A0[0], v0, v0); else fm1_run<I>(A0[I], v0, v0);      // (i, j)      += v_i v_j
        if constexpr (16 + I < DR) {
            fm1_run<I>(A0[16 + I], v1, v0);                                               // (16+i, j)   += v_(16+i) v_j
            fm1_run<I>(A1[16 + I], v1, v1);                                               // (16+i,16+j) += v_(16+i) v_(16+j)
        }
        col_rank1<DR, I + 1>(A0, A1, v0, v1);
    }
}
// rows I .. DR-1 and the extra row 32 of step k = 16 S + K: A_s[I] -= A[I][k] * A[k][c_s] / d_k
template <int DR, int S, int K, int I, bool FIRST>
__device__ __forceinline__ void fin_elim(double (&A0)[33], double (&A1)[33], double nm0, double nm1)
{
    if constexpr (I <= 32) {
        if constexpr (S == 0) {
            if constexpr (DR > 16) {
                if constexpr (FIRST) fm1<K>(A1[I], A0[I], nm1); else fm1_run<K>(A1[I], A0[I], nm1);
                fm1_self_run<K>(A0[I], nm0);
            } else {
                if constexpr (FIRST) fm1_self<K>(A0[I], nm0); else fm1_self_run<K>(A0[I], nm0);
            }
        } else {
            if constexpr (FIRST) fm1_self<K>(A1[I], nm1); else fm1_self_run<K>(A1[I], nm1);
        }
        fin_elim<DR, S, K, (I + 1 < DR || I == 32) ? I + 1 : 32, false>(A0, A1, nm0, nm1);
    }
}

template <int DR, int k>
__device__ __forceinline__ void fin_factor(double (&A0)[33], double (&A1)[33], double &d0, double &d1, int j)
{
    if constexpr (k < DR) {
        constexpr int S = k / 16, K = k % 16;
        const double dk = row_bcast_f64<K>(S ? A1[k] : A0[k]);
        const double rinv = fast_rcp(dk);
        double nm0 = 0.0, nm1;
        if constexpr (S == 0) {
            d0 = (j == K) ? dk : d0;
            nm0 = (j > K) ? -(A0[k] * rinv) : 0.0;             // finished columns are left alone
            nm1 = -(A1[k] * rinv);
        } else {
            d1 = (j == K) ? dk : d1;
            nm1 = (j > K) ? -(A1[k] * rinv) : 0.0;
        }
        fin_elim<DR, S, K, (k + 1 < DR) ? k + 1 : 32, true>(A0, A1, nm0, nm1);
        fin_factor<DR, k + 1>(A0, A1, d0, d1, j);
    }
}

// y_c -= l_Cc x_C for the columns left of C, C = DR-1 .. 1 (x_C is final in lane C % 16 when its turn comes)
template <int C>
__device__ __forceinline__ void fin_backward(const double (&A0)[33], const double (&A1)[33], double &y0, double &y1, double rd0,
                                             double rd1, int j)
{
    if constexpr (C >= 1) {
        constexpr int S = C / 16, K = C % 16;
        if constexpr (S == 1) {
            fm1<K>(y0, y1, -(A0[C] * rd0));
            if constexpr (K > 0) fm1_self_run<K>(y1, (j < K) ? -(A1[C] * rd1) : 0.0);
        } else {
            fm1_self<K>(y0, (j < K) ? -(A0[C] * rd0) : 0.0);
        }
        fin_backward<C - 1>(A0, A1, y0, y1, rd0, rd1, j);
    }
}

}  // namespace
