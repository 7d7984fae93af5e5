// dpp_rows16.h -- a 16 x 16 (or smaller) symmetric positive-definite system per ROW OF 16 LANES, four systems per wavefront:
// lane j of a lane row holds COLUMN j (A[i] = entry (i, j)) and entry j of the right-hand side.  Rank-1 updates, the LDL'
// factorisation with the forward solve riding along, and the backward solve are v_fmac_f64_dpp row_newbcast instructions
// (lane k of each lane row is the broadcast source).  Shared by k_rows_small (k_sample_rows.hip: D <= 16, four entity rows per
// wave) and k_rows_lr4 (k_rows_lr.hip: the n x n systems of the low-rank sampler).
#pragma once
#include "c_layout_chol.h"

namespace {

template <int K>
__device__ __forceinline__ uint32_t row_bcast_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + K, 0xf, 0xf, false);      // row_newbcast:K
}
template <int K>
__device__ __forceinline__ double row_bcast_f64(double v)
{
    double o;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "n"(K));
    return o;
}

// (DPP reads of a register need two wait states after a vector instruction wrote it: the s_nop in fm1 / fm1_self.  In the
// runs below the DPP source was written long before -- the gathered value once per observation, a matrix row in the step
// before -- so only the first instruction of a run carries the s_nop)
template <int KJ>
__device__ __forceinline__ void fm1_run(double &d, double s, double m)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s), "v"(m), "n"(KJ));
}
template <int KJ>
__device__ __forceinline__ void fm1_self_run(double &d, double m)
{
    asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(m), "n"(KJ));
}
// DR: D rounded up to a multiple of four -- rows DR .. 15 of the padded system are never touched
template <int DR, int I>
__device__ __forceinline__ void small_rank1(double (&A)[16], double v)
{
    if constexpr (I < DR) {
        if constexpr (I == 0) fm1<I>(A[I], v, v); else fm1_run<I>(A[I], v, v);       // A[I][j] += v_I v_j
        small_rank1<DR, I + 1>(A, v);
    }
}
template <int DR, int K, int I>
__device__ __forceinline__ void small_elim(double (&A)[16], double nm)
{
    if constexpr (I < DR) {
        if constexpr (I == K + 1) fm1_self<K>(A[I], nm); else fm1_self_run<K>(A[I], nm);      // A[I][j] -= A[I][K] m_j
        small_elim<DR, K, I + 1>(A, nm);
    }
}
// (no branches around the DPP instructions: steps of the padding -- rows and columns D .. DR-1 are the identity -- are
// no-ops by their values, and a branch costs the copies the compiler makes for the asm operands at every merge)
template <int DR, int K>
__device__ __forceinline__ void small_factor(double (&A)[16], double &b, double &dj, int j)
{
    if constexpr (K < DR) {
        const double dk = row_bcast_f64<K>(A[K]);
        dj = (j == K) ? dk : dj;
        const double nm = (j > K) ? -(A[K] * fast_rcp(dk)) : 0.0;            // -A[K][j] / d_K; finished columns are left alone
        small_elim<DR, K, K + 1>(A, nm);
        fm1_self<K>(b, nm);                                                // the forward solve: b_j -= w_K l_jK
        small_factor<DR, K + 1>(A, b, dj, j);
    }
}
template <int C>
__device__ __forceinline__ void small_backward(const double (&A)[16], double &y, double rdj, int j)
{
    if constexpr (C >= 1) {
        fm1_self<C>(y, (j < C) ? -(A[C] * rdj) : 0.0);                     // y_j -= l_Cj x_C for the columns left of C
        small_backward<C - 1>(A, y, rdj, j);
    }
}
}  // namespace
