// k_feat.hip -- K2..K5: the side-information (Entity.F) operators and the beta update.
//
//   Entity.F operator contract: F*B, At_mul_B(F,B), AtA_mul_B! for dense F, SparseMatrixCSR
//   (src/parallel_csr.jl:36-54) and binary sparse F (src/sparsebin_csr.jl:22-63, src/parallel_matrix.jl:19-24,
//   242-267).  Sparse operators keep the CSR of F and the CSR of F' so that both products are row gathers with
//   no atomics (the reference's COO At_mul_B! scatters, parallel_matrix.jl:258-267).
//   uhat = (F beta)'                                   F_mul_beta, src/RelationData.jl:314-320; macau.jl:103,112
//   rhs  = F'((sample - mu)' + E1) + sqrt(lb) E2       sample_beta, src/sampling.jl:298-300
//   beta = (F'F + lb I) \ rhs                          solve_full :314-320 | solve_cg2 parallel_matrix.jl:488-507
//   all D conjugate-gradient solves advance together, each column keeping the reference's own stopping rule
//   (cg_AtA, src/parallel_cg.jl:63-94: stop when ||r|| < tol ||b||, checked before an iteration; maxiter).
//   lambda_beta ~ Gamma                                sample_lambda_beta, src/sampling.jl:136-142
#include "bdf_common.h"
#include <chrono>
#include "wave_linalg.h"
#include "dpp_rows16.h"
#include <algorithm>
#include <cmath>

namespace {

// ---- strided dense GEMM: C(i,j) = sum_k A(i,k) B(k,j), optional second output C2 = C + bias[j] -------------
constexpr int TM = 32, TN = 32, TK = 16;

struct GemmArgs {
    int64_t M, N, K;
    const double *A; int64_t ars, acs;
    const double *B; int64_t brs, bcs;
    double *C; int64_t crs, ccs;
    const double *bias; double *C2;
};

// blockIdx.z = K chunk (split-K): with more than one chunk the tile goes to part[z][i][j] (M x N row-major per chunk) and
// k_gemm_reduce adds the chunks in order -- a tall-and-skinny F' T (K = rows of F) would otherwise run on a handful of CUs
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g, int64_t kchunk, double *part, const int *skip)
{
    if (skip && *skip == 0) return;
    __shared__ double As[TK][TM + 1];
    __shared__ double Bs[TK][TN + 1];
    const int tid = threadIdx.x;
    const int tx = tid % 16, ty = tid / 16;
    const int64_t i0 = (int64_t)blockIdx.x * TM, j0 = (int64_t)blockIdx.y * TN;
    const int64_t kb = (int64_t)blockIdx.z * kchunk, ke = (kb + kchunk < g.K) ? kb + kchunk : g.K;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    const bool a_fast_i = g.ars <= g.acs, b_fast_k = g.brs <= g.bcs;
    for (int64_t k0 = kb; k0 < ke; k0 += TK) {
        for (int e = tid; e < TM * TK; e += 256) {
            const int ii = a_fast_i ? e % TM : e / TK, kk = a_fast_i ? e / TM : e % TK;
            const int64_t i = i0 + ii, k = k0 + kk;
            As[kk][ii] = (i < g.M && k < ke) ? g.A[i * g.ars + k * g.acs] : 0.0;
        }
        for (int e = tid; e < TN * TK; e += 256) {
            const int kk = b_fast_k ? e % TK : e / TN, jj = b_fast_k ? e / TK : e % TN;
            const int64_t k = k0 + kk, j = j0 + jj;
            Bs[kk][jj] = (k < ke && j < g.N) ? g.B[k * g.brs + j * g.bcs] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; kk++) {
            const double a0 = As[kk][tx], a1 = As[kk][tx + 16];
            const double b0 = Bs[kk][ty], b1 = Bs[kk][ty + 16];
            acc[0][0] = fma(a0, b0, acc[0][0]); acc[0][1] = fma(a0, b1, acc[0][1]);
            acc[1][0] = fma(a1, b0, acc[1][0]); acc[1][1] = fma(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const int64_t i = i0 + tx + 16 * u, j = j0 + ty + 16 * v;
            if (i < g.M && j < g.N) {
                if (part) {
                    part[((int64_t)blockIdx.z * g.M + i) * g.N + j] = acc[u][v];
                } else {
                    g.C[i * g.crs + j * g.ccs] = acc[u][v];
                    if (g.C2) g.C2[i * g.crs + j * g.ccs] = acc[u][v] + g.bias[j];
                }
            }
        }
}

__global__ __launch_bounds__(256) void k_gemm_reduce(GemmArgs g, int nchunks, const double *part, const int *skip)
{
    if (skip && *skip == 0) return;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= g.M * g.N) return;
    const int64_t i = e / g.N, j = e % g.N;
    double s = 0.0;
    for (int z = 0; z < nchunks; z++) s += part[(int64_t)z * g.M * g.N + e];       // fixed order
    g.C[i * g.crs + j * g.ccs] = s;
    if (g.C2) g.C2[i * g.crs + j * g.ccs] = s + g.bias[j];
}

int gemm(bdf_ctx *ctx, const GemmArgs &g)
{
    if (g.M == 0 || g.N == 0) return BDF_OK;
    const int64_t tiles = ((g.M + TM - 1) / TM) * ((g.N + TN - 1) / TN);
    int nchunks = 1;
    if (tiles < 512 && g.K >= 128) {                       // too few tiles to fill the chip and a long K: split it
        nchunks = (int)std::min<int64_t>(64, std::min<int64_t>((g.K + 63) / 64, (1024 + tiles - 1) / tiles));
        if (nchunks < 1) nchunks = 1;
    }
    dim3 grid((unsigned)((g.M + TM - 1) / TM), (unsigned)((g.N + TN - 1) / TN), (unsigned)nchunks);
    if (nchunks == 1) {
        hipLaunchKernelGGL(k_gemm, grid, dim3(256), 0, ctx->stream, g, g.K, (double *)nullptr, ctx->skip_flag);
    } else {
        void *sc;
        int rc = bdf_scratch2(ctx, (size_t)nchunks * g.M * g.N * sizeof(double), &sc);
        if (rc) return rc;
        const int64_t kchunk = ((g.K + nchunks - 1) / nchunks + TK - 1) / TK * TK;
        hipLaunchKernelGGL(k_gemm, grid, dim3(256), 0, ctx->stream, g, kchunk, (double *)sc, ctx->skip_flag);
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((g.M * g.N + 255) / 256)), dim3(256), 0, ctx->stream, g, nchunks,
                           (const double *)sc, ctx->skip_flag);
    }
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

// ---- dense feature matrices on the matrix cores (v_mfma_f64_16x16x4_f64), at most 64 right-hand columns --------------
// F is N x numF column-major.  These are the two genuinely dense contractions of the path (F beta and F' T over
// the 24 MB of a 6040 x 500 F, AI ~ 8 flop/B per pass).
typedef double fd4 __attribute__((ext_vector_type(4)));

// Y(r, c) = sum_k F(r, k) B(k, c) for a column-major F: one wave per 16 rows x all columns, every operand straight from
// global memory in the MFMA's lane layout, no LDS, no barrier.  Lane (i = l & 15, h = l >> 4) supplies F(row i, k) -- 16
// consecutive rows of a column are one 128-byte read -- and B(k, column i).  Which k of a 16-chunk a lane takes in MFMA
// step t is free as long as A and B agree: k = 4h + t when B is column-major (the lane's four values of B are then 32
// contiguous bytes), k = 4t + h otherwise (the 16 lanes of an h read 128 contiguous bytes of a row of B).  B is small
// (K x ncol) and shared by all waves: it stays in L1/L2.  The chunk after the current one is loaded before the current
// one's MFMAs.  The four waves of a workgroup share the 16 rows and split K (a 500 x 500 F'F has only 32 row tiles), wave
// 0 adds their results in wave order.  (A version that staged B through LDS for four waves of different rows took 36 us
// for the 6040 x 500 x 32 product and as long for the 500 x 500 x 32 one: 95 resp. 8 workgroups, two barriers per 64 k.)
template <int CB, bool CM>
__global__ __launch_bounds__(256) void k_dense_nn(const double *__restrict__ F, int64_t M, int64_t K, const double *__restrict__ B,
                                                 int64_t brs, int64_t bcs, int ncol, double *__restrict__ Y, int64_t yrs,
                                                 int64_t ycs, const double *__restrict__ bias, double *__restrict__ Y2,
                                                 const int *skip)
{
    __shared__ double red[3][CB][4][64];
    if (skip && *skip == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, h = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    const int64_t row = r0 + i;
    const bool rok = row < M;
    const int64_t kq = ((K + 3) / 4 + 15) / 16 * 16;      // this wave's K range: [kb, ke)
    const int64_t kb = wave * kq, ke = (kb + kq < K) ? kb + kq : K;
    fd4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) acc[cb] = fd4{0.0, 0.0, 0.0, 0.0};
    double a[2][4], b[2][CB][4];
    auto load = [&](int64_t k0, int S) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int64_t k = k0 + (CM ? 4 * h + t : 4 * t + h);
            const bool kok = k < ke;
            a[S][t] = (rok && kok) ? F[row + k * M] : 0.0;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const int c = 16 * cb + i;
                b[S][cb][t] = (kok && c < ncol) ? B[k * brs + (int64_t)c * bcs] : 0.0;
            }
        }
    };
    load(kb, 0);
    for (int64_t k0 = kb; k0 < ke; k0 += 32) {
        load(k0 + 16, 1);                               // beyond the range: zeros
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int cb = 0; cb < CB; cb++) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][t], b[0][cb][t], acc[cb], 0, 0, 0);
        load(k0 + 32, 0);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int cb = 0; cb < CB; cb++) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][t], b[1][cb][t], acc[cb], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[wave - 1][cb][r][lane] = acc[cb][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int cb = 0; cb < CB; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double v = ((acc[cb][r] + red[0][cb][r][lane]) + red[1][cb][r][lane]) + red[2][cb][r][lane];
            const int64_t rr = r0 + h + 4 * r;
            const int c = 16 * cb + i;
            if (rr < M && c < ncol) {
                Y[rr * yrs + (int64_t)c * ycs] = v;
                if (Y2) Y2[rr * yrs + (int64_t)c * ycs] = v + bias[c];
            }
        }
}

// part[z][f][c] = sum over the rows of chunk z of F(row, f) B(row, c)   (= F' B by chunks; k_gemm_reduce adds the chunks
// in order).  A workgroup owns 16 features and one row chunk; its waves take 64-row tiles in turn: the F tile goes through
// LDS (read along the rows, 512 contiguous bytes per feature; the MFMA wants feature-major), B operands from global.
template <int CB>
__global__ __launch_bounds__(256) void k_dense_tn(const double *__restrict__ F, int64_t M, int64_t numF, const double *__restrict__ B,
                                                  int64_t brs, int64_t bcs, int ncol, int64_t rows_per_chunk,
                                                  double *__restrict__ part, const int *skip)
{
    __shared__ double tile[4][16][65];
    if (skip && *skip == 0) return;
    __shared__ double red[3][CB][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, h = lane >> 4;
    const int64_t f0 = (int64_t)blockIdx.x * 16;
    const int64_t c0 = (int64_t)blockIdx.y * rows_per_chunk, c1 = (c0 + rows_per_chunk < M) ? c0 + rows_per_chunk : M;
    fd4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) acc[cb] = fd4{0.0, 0.0, 0.0, 0.0};
    for (int64_t rr = c0 + 64 * wave; rr < c1; rr += 256) {
        const int64_t myrow = rr + lane;
#pragma unroll
        for (int ff = 0; ff < 16; ff++)
            tile[wave][ff][lane] = (myrow < c1 && f0 + ff < numF) ? F[myrow + (f0 + ff) * M] : 0.0;
        double b[16][CB];
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const int64_t row = rr + 4 * s + h;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const int c = 16 * cb + i;
                b[s][cb] = (row < c1 && c < ncol) ? B[row * brs + (int64_t)c * bcs] : 0.0;
            }
        }
        wave_sync();
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const double a = tile[wave][i][4 * s + h];
#pragma unroll
            for (int cb = 0; cb < CB; cb++) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s][cb], acc[cb], 0, 0, 0);
        }
        wave_sync();
    }
    if (wave > 0)
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[wave - 1][cb][r][lane] = acc[cb][r];
    __syncthreads();
    if (wave == 0) {
        double *p = part + (int64_t)blockIdx.y * numF * ncol;
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double v = acc[cb][r];
#pragma unroll
                for (int w = 0; w < 3; w++) v += red[w][cb][r][lane];
                const int64_t ff = f0 + h + 4 * r;
                const int c = 16 * cb + i;
                if (ff < numF && c < ncol) p[ff * ncol + c] = v;
            }
    }
}

// Y = A B for a dense column-major M x K matrix A (a feature matrix, or the precomputed F'F), ncol <= 64
int dense_nn(bdf_ctx *ctx, const double *A, int64_t M, int64_t K, const double *B, int64_t brs, int64_t bcs, int ncol,
             double *Y, int64_t yrs, int64_t ycs, const double *bias, double *Y2)
{
    const int CB = (ncol + 15) / 16;
    const bool cm = brs == 1 && bcs != 1;         // column-major B
    dim3 grid((unsigned)((M + 15) / 16));
#define NN(C) do { if (cm) hipLaunchKernelGGL((k_dense_nn<C, true>), grid, dim3(256), 0, ctx->stream, A, M, K, B, brs, bcs, ncol, Y, yrs, ycs, bias, Y2, ctx->skip_flag); \
                   else hipLaunchKernelGGL((k_dense_nn<C, false>), grid, dim3(256), 0, ctx->stream, A, M, K, B, brs, bcs, ncol, Y, yrs, ycs, bias, Y2, ctx->skip_flag); } while (0)
    if (CB == 1) NN(1); else if (CB == 2) NN(2); else if (CB == 3) NN(3); else NN(4);
#undef NN
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

#ifndef BDF_TN_WGS
#define BDF_TN_WGS 1024       // workgroups the F' B product aims at (feature tiles x row chunks)
#endif
int dense_apply(bdf_ctx *ctx, const bdf_feat *f, bool transpose, const double *B, int64_t brs, int64_t bcs, int ncol,
                double *Y, int64_t yrs, int64_t ycs, const double *bias, double *Y2)
{
    const int CB = (ncol + 15) / 16;
    if (!transpose) return dense_nn(ctx, f->dense_dev, f->m, f->n, B, brs, bcs, ncol, Y, yrs, ycs, bias, Y2);
    const int64_t ftiles = (f->n + 15) / 16;
    int64_t nchunks = std::max<int64_t>(1, std::min<int64_t>((f->m + 255) / 256, (BDF_TN_WGS + ftiles - 1) / ftiles));
    const int64_t rpc = ((f->m + nchunks - 1) / nchunks + 63) / 64 * 64;
    nchunks = (f->m + rpc - 1) / rpc;
    void *sc;
    int rc = bdf_scratch2(ctx, (size_t)nchunks * f->n * ncol * sizeof(double), &sc);
    if (rc) return rc;
    dim3 grid((unsigned)ftiles, (unsigned)nchunks);
#define TN(C) hipLaunchKernelGGL(k_dense_tn<C>, grid, dim3(256), 0, ctx->stream, (const double *)f->dense_dev, f->m, f->n, B, brs, bcs, ncol, rpc, (double *)sc, ctx->skip_flag)
    if (CB == 1) TN(1); else if (CB == 2) TN(2); else if (CB == 3) TN(3); else TN(4);
#undef TN
    GemmArgs g;
    g.M = f->n; g.N = ncol; g.K = f->m; g.A = nullptr; g.ars = g.acs = 0; g.B = nullptr; g.brs = g.bcs = 0;
    g.C = Y; g.crs = yrs; g.ccs = ycs; g.bias = bias; g.C2 = Y2;
    hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((g.M * g.N + 255) / 256)), dim3(256), 0, ctx->stream, g, (int)nchunks,
                       (const double *)sc, ctx->skip_flag);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

// ---- sparse (CSR) x dense: Y(r,c) = sum_q val_q B(col_q, c); vals == NULL means implicit 1.0 ------------------
// The kernel wants both dense operands ROW-major (a gathered row of B is then one contiguous read of 8 ncol bytes shared
// by the lanes that walk the columns; with a column-major B every nonzero touches ncol different cache lines: measured
// 2.0 ms per product on config C5's 100,000 x 50,000 binary matrix, 5M nonzeros, 32 columns, against ~0.3 ms).  Operands in
// another layout -- the CG state and beta are column-major like the reference's matrices -- pass through a tiled transpose
// into / out of scratch.
struct SpmmArgs {
    int64_t m, kin; int ncol;        // m rows of the sparse operand (outputs), kin rows of B
    const int64_t *rowptr; const int32_t *colind; const double *vals;
    const double *B; int64_t brs, bcs;
    double *Y; int64_t yrs, ycs;
    const double *bias; double *Y2;
    const int64_t *panel_ptr = nullptr; int n_panels = 0;     // column panels of the sparse operand (bdf_feat::panel_*), or none
};

// rows of B per column panel: 3 MiB of a 32-column operand (a 4 MiB XCD L2 keeps the panel beside the streams of the launch)
#define BDF_SPMM_PANEL_ROWS 12288

// B(i,c) at B[i*ldb + c], Y(r,c) at Y[r*ldy + c].  32 lanes walk the columns, 8 rows per block; the row's nonzeros four at a
// time (independent gathers), accumulated in order (the result does not depend on the unrolling).
__global__ __launch_bounds__(256) void k_spmm_rm(int64_t m, int ncol, const int64_t *__restrict__ rowptr,
                                                 const int32_t *__restrict__ colind, const double *__restrict__ vals,
                                                 const double *__restrict__ B, int64_t ldb, double *__restrict__ Y, int64_t ldy,
                                                 const double *__restrict__ bias, double *__restrict__ Y2, const int *skip)
{
    if (skip && *skip == 0) return;
    const int c0 = threadIdx.x % 32;
    const int64_t r = (int64_t)blockIdx.x * 8 + threadIdx.x / 32;
    if (r >= m) return;
    const int64_t beg = rowptr[r], end = rowptr[r + 1];
    for (int c = c0; c < ncol; c += 32) {
        double acc = 0.0;
        int64_t q = beg;
        for (; q + 4 <= end; q += 4) {
            const int32_t i0 = colind[q], i1 = colind[q + 1], i2 = colind[q + 2], i3 = colind[q + 3];
            const double b0 = B[(int64_t)i0 * ldb + c], b1 = B[(int64_t)i1 * ldb + c], b2 = B[(int64_t)i2 * ldb + c],
                         b3 = B[(int64_t)i3 * ldb + c];
            if (vals) {
                acc = fma(vals[q], b0, acc); acc = fma(vals[q + 1], b1, acc);
                acc = fma(vals[q + 2], b2, acc); acc = fma(vals[q + 3], b3, acc);
            } else {
                acc = fma(1.0, b0, acc); acc = fma(1.0, b1, acc); acc = fma(1.0, b2, acc); acc = fma(1.0, b3, acc);
            }
        }
        for (; q < end; q++) acc = fma(vals ? vals[q] : 1.0, B[(int64_t)colind[q] * ldb + c], acc);
        Y[r * ldy + c] = acc;
        if (Y2) Y2[r * ldy + c] = acc + bias[c];
    }
}

// The same product for up to 32 columns taken in pairs (C5: D = 32, 5 M nonzeros; the kernel above ran 149 us per product there,
// 0.05 of what its bytes cost at the HBM rate -- four 8-byte gathers in flight per lane, one dependent round trip after the
// other).  SIXTEEN lanes per row, 16 bytes per lane (a 256-byte row of B is one instruction of the lane row), four rows per wave,
// sixteen rows per workgroup; a row's column indices come sixteen at a time -- lane l of the lane row loads index l of the chunk,
// one coalesced read -- and are handed round by DPP row broadcasts, and all (up to) sixteen gathers of a chunk are issued before
// the first is used: 256 bytes x 16 x 4 rows = 16 KB in flight per wave.  Same sums in the same order as k_spmm_rm (entry q of
// the row after entry q - 1): the two kernels agree to the last bit.
typedef double spd2 __attribute__((ext_vector_type(2)));
template <int J>
__device__ __forceinline__ void spmm_gather16(spd2 (&g)[16], int32_t myi, int left, const double *__restrict__ B, int64_t ldb, int c, bool cv)
{
    if constexpr (J < 16) {
        const int32_t ij = (int32_t)row_bcast_u32<J>((uint32_t)myi);
        g[J] = (cv && J < left) ? *(const spd2 *)(B + (int64_t)ij * ldb + c) : spd2{0.0, 0.0};
        spmm_gather16<J + 1>(g, myi, left, B, ldb, c, cv);
    }
}
template <bool HASV, int J>
__device__ __forceinline__ void spmm_acc16(const spd2 (&g)[16], double myv, int left, double &a0, double &a1)
{
    if constexpr (J < 16) {
        const double w = HASV ? row_bcast_f64<J>(myv) : (J < left ? 1.0 : 0.0);
        a0 = fma(w, g[J][0], a0);
        a1 = fma(w, g[J][1], a1);
        spmm_acc16<HASV, J + 1>(g, myv, left, a0, a1);
    }
}
template <bool HASV>
__global__ __launch_bounds__(256) void k_spmm_rm16(int64_t m, int ncol, const int64_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ colind, const double *__restrict__ vals,
                                                   const double *__restrict__ B, int64_t ldb, double *__restrict__ Y, int64_t ldy,
                                                   const double *__restrict__ bias, double *__restrict__ Y2, const int *skip,
                                                   const int64_t *__restrict__ pb, const int64_t *__restrict__ pe, int first)
{
    if (skip && *skip == 0) return;
    const int l = threadIdx.x & 15;
    const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool rv = r < m;                       // (every lane stays: the broadcasts run over whole lane rows)
    // pb / pe: this launch takes the row's entries [pb[r], pe[r]) -- one column panel of the operand -- and carries the row's running
    // sums on from Y unless it is the first panel (the entries of a row in column order: the same sums in the same order as one pass)
    const int64_t beg = rv ? (pb ? pb[r] : rowptr[r]) : 0, end = rv ? (pb ? pe[r] : rowptr[r + 1]) : 0;
    const int c = 2 * l;
    const bool cv = c < ncol;
    double a0 = 0.0, a1 = 0.0;
    if (!first && rv && cv) { const spd2 y = *(const spd2 *)(Y + r * ldy + c); a0 = y[0]; a1 = y[1]; }
    // (the longest row of the wave sets the trip count: wave-uniform, the DPP instructions never sit under a divergent branch)
    int64_t nq = end - beg;
    nq = max(nq, __shfl_xor(nq, 16));
    nq = max(nq, __shfl_xor(nq, 32));
    nq = __builtin_amdgcn_readfirstlane((int)nq);
    // (the NEXT chunk's indices -- and values -- are loaded before this chunk's gathers are issued: a chunk then costs one dependent
    // round trip, its gathers, instead of two)
    int32_t myi = beg + l < end ? colind[beg + l] : 0;
    double myv = 0.0;
    if (HASV) myv = beg + l < end ? vals[beg + l] : 0.0;
    for (int64_t o = 0; o < nq; o += 16) {
        const int left = (int)min((int64_t)16, end - beg - o);          // entries of this lane row's chunk (<= 0: none)
        const int64_t qn = beg + o + 16 + l;
        const int32_t nxi = qn < end ? colind[qn] : 0;
        double nxv = 0.0;
        if (HASV) nxv = qn < end ? vals[qn] : 0.0;
        spd2 g[16];
        spmm_gather16<0>(g, myi, left, B, ldb, c, cv);
        spmm_acc16<HASV, 0>(g, myv, left, a0, a1);
        myi = nxi; myv = nxv;
    }
    if (rv && cv) {
        *(spd2 *)(Y + r * ldy + c) = spd2{a0, a1};
        if (Y2) *(spd2 *)(Y2 + r * ldy + c) = spd2{a0 + bias[c], a1 + bias[c + 1]};
    }
}

// The column panels of a product in ONE launch (round 6; until then one launch of the kernel above per panel, the rows' running
// sums carried through Y: 5 + 9 launches per F'(F p) on configuration C5, and Y -- 25.6 MB -- written and read back between them).
// A PERSISTENT grid, one workgroup per resident slot: workgroup w owns the row blocks w, w + G, ... (KB of them, sixteen rows
// each), walks the panels in order and inside a panel its row blocks, and keeps every row's two running sums in registers from the
// first panel to the last -- Y is written once.  Nothing synchronises the workgroups: they start together and do the same amount of
// work per panel, so the chip is inside one panel (two at the edges) at any moment and every XCD's L2 holds the 3 MiB of the operand
// its gathers want.  A row's entries are taken in the order of k_spmm_rm16 (column order: panel after panel): the same sums to the
// last bit.  The walk is pipelined over the units (row block, panel): the unit after the next one's bounds and the next one's
// first sixteen indices are loaded before this unit's gathers are issued -- a unit of ~10 entries per row is ONE dependent round
// trip, its gathers.
template <bool HASV>
__global__ __launch_bounds__(256, 4) void k_spmm_rm16p(int64_t m, int ncol, const int32_t *__restrict__ colind, const double *__restrict__ vals,
                                                       const double *__restrict__ B, int64_t ldb, double *__restrict__ Y, int64_t ldy,
                                                       const double *__restrict__ bias, double *__restrict__ Y2, const int *skip,
                                                       const int64_t *__restrict__ panel_ptr, int np, int KB, int64_t rb0, int64_t nblocks)
{
    // the rows' running sums: KB pairs per thread, in LDS (in registers they cost the kernel its fourth wave per SIMD -- and a grid
    // sized for four that holds three runs its last quarter as a second generation, out of step with the panels)
    extern __shared__ __attribute__((aligned(16))) double spmm_acc[];
    // (A counter the workgroups add to after every panel and briefly wait on was tried as a hint to keep them in step: its ~900
    // pollers on one word starve the arrivals -- every wait ran into its bound, 510 us per product instead of 120.  Not kept.)
    if (skip && *skip == 0) return;
    const int l = threadIdx.x & 15;
    const int c = 2 * l;
    const bool cv = c < ncol;
    const int64_t G = gridDim.x;
    spd2 *acc = (spd2 *)spmm_acc + threadIdx.x;                   // pair k of this thread: acc[k * 256]
    for (int k = 0; k < KB; k++) acc[k * 256] = spd2{0.0, 0.0};
    // unit u = p * KB + k: row block rb0 + blockIdx.x + k G, panel p
    auto row_of = [&](int k) -> int64_t {
        const int64_t rb = rb0 + blockIdx.x + (int64_t)k * G;
        const int64_t r = rb * 16 + (threadIdx.x >> 4);
        return (rb < nblocks && r < m) ? r : -1;
    };
    const int n_units = np * KB;
    // the pipeline's registers: bounds two units ahead, bounds + first indices one unit ahead
    int64_t b2 = 0, e2 = 0, b1 = 0, e1 = 0;
    int32_t i1 = 0;
    double v1 = 0.0;
    int k2 = 0, p2 = 0;                                           // (k, p) of the unit whose bounds are loaded next
    auto bounds = [&](int64_t &b, int64_t &e) {
        b = e = 0;
        if (p2 < np) {
            const int64_t r = row_of(k2);
            if (r >= 0) { const int64_t *pp = panel_ptr + (int64_t)p2 * m + r; b = pp[0]; e = pp[m]; }
        }
        if (++k2 == KB) { k2 = 0; p2++; }
    };
    bounds(b1, e1);
    bounds(b2, e2);
    i1 = b1 + l < e1 ? colind[b1 + l] : 0;
    if (HASV) v1 = b1 + l < e1 ? vals[b1 + l] : 0.0;
    (void)n_units;
#pragma unroll 1
    for (int p = 0; p < np; p++) {
#pragma unroll 1
        for (int k = 0; k < KB; k++) {
            const int64_t beg = b1, end = e1;
            int32_t myi = i1;
            double myv = v1;
            // the next unit's first indices and the one after's bounds: in flight under this unit's gathers
            b1 = b2; e1 = e2;
            i1 = b1 + l < e1 ? colind[b1 + l] : 0;
            if (HASV) v1 = b1 + l < e1 ? vals[b1 + l] : 0.0;
            bounds(b2, e2);
            int64_t nq = end - beg;
            nq = max(nq, __shfl_xor(nq, 16));
            nq = max(nq, __shfl_xor(nq, 32));
            nq = __builtin_amdgcn_readfirstlane((int)nq);
            if (nq <= 0) continue;
            spd2 av = acc[k * 256];
            double a0 = av[0], a1 = av[1];
            for (int64_t o = 0; o < nq; o += 16) {
                const int left = (int)min((int64_t)16, end - beg - o);
                const int64_t qn = beg + o + 16 + l;
                int32_t nxi = 0;
                double nxv = 0.0;
                if (o + 16 < nq) {                                   // (rare: a row with more than sixteen entries in one panel)
                    nxi = qn < end ? colind[qn] : 0;
                    if (HASV) nxv = qn < end ? vals[qn] : 0.0;
                }
                spd2 g[16];
                spmm_gather16<0>(g, myi, left, B, ldb, c, cv);
                spmm_acc16<HASV, 0>(g, myv, left, a0, a1);
                myi = nxi; myv = nxv;
            }
            acc[k * 256] = spd2{a0, a1};
        }
    }
    for (int k = 0; k < KB; k++) {
        const int64_t r = row_of(k);
        if (r >= 0 && cv) {
            const spd2 av = acc[k * 256];
            *(spd2 *)(Y + r * ldy + c) = av;
            if (Y2) *(spd2 *)(Y2 + r * ldy + c) = spd2{av[0] + bias[c], av[1] + bias[c + 1]};
        }
    }
}

// out[i*ncol + c] = in[i*irs + c*ics]  (32 x 32 tiles through LDS: coalesced on both sides for a column-major `in`)
__global__ __launch_bounds__(256) void k_to_rowmajor(int64_t n, int ncol, const double *__restrict__ in, int64_t irs, int64_t ics,
                                                     double *__restrict__ out, const int *skip)
{
    __shared__ double t[32][33];
    if (skip && *skip == 0) return;
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32, a = threadIdx.x % 32;
    for (int b = threadIdx.x / 32; b < 32; b += 8) {
        const int64_t i = i0 + a;
        const int c = c0 + b;
        t[b][a] = (i < n && c < ncol) ? in[i * irs + c * ics] : 0.0;
    }
    __syncthreads();
    for (int b = threadIdx.x / 32; b < 32; b += 8) {
        const int64_t i = i0 + b;
        const int c = c0 + a;
        if (i < n && c < ncol) out[i * ncol + c] = t[a][b];
    }
}

// out[i*ors + c*ocs] = in[i*ncol + c]  (+ the biased copy out2)
__global__ __launch_bounds__(256) void k_from_rowmajor(int64_t n, int ncol, const double *__restrict__ in, double *__restrict__ out,
                                                       int64_t ors, int64_t ocs, const double *__restrict__ bias,
                                                       double *__restrict__ out2, const int *skip)
{
    __shared__ double t[32][33];
    if (skip && *skip == 0) return;
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32, a = threadIdx.x % 32;
    for (int b = threadIdx.x / 32; b < 32; b += 8) {
        const int64_t i = i0 + b;
        const int c = c0 + a;
        t[b][a] = (i < n && c < ncol) ? in[i * ncol + c] : 0.0;
    }
    __syncthreads();
    for (int b = threadIdx.x / 32; b < 32; b += 8) {
        const int64_t i = i0 + a;
        const int c = c0 + b;
        if (i < n && c < ncol) {
            const double v = t[a][b];
            out[i * ors + c * ocs] = v;
            if (out2) out2[i * ors + c * ocs] = v + bias[c];
        }
    }
}

int spmm(bdf_ctx *ctx, const SpmmArgs &s)
{
    if (s.m == 0 || s.ncol == 0) return BDF_OK;
    const bool b_rm = s.bcs == 1 || s.ncol == 1, y_rm = s.ycs == 1 || s.ncol == 1;
    const double *B = s.B;
    int64_t ldb = s.brs;
    double *Y = s.Y;
    int64_t ldy = s.yrs;
    if (!b_rm || !y_rm) {
        void *sc;
        int rc = bdf_scratch2(ctx, (size_t)((b_rm ? 0 : s.kin) + (y_rm ? 0 : s.m)) * s.ncol * sizeof(double), &sc);
        if (rc) return rc;
        double *tb = (double *)sc, *ty = (double *)sc + (b_rm ? 0 : s.kin * s.ncol);
        if (!b_rm) {
            if (s.kin > 0)
                hipLaunchKernelGGL(k_to_rowmajor, dim3((unsigned)((s.kin + 31) / 32), (unsigned)((s.ncol + 31) / 32)), dim3(256), 0,
                                   ctx->stream, s.kin, s.ncol, s.B, s.brs, s.bcs, tb, ctx->skip_flag);
            B = tb; ldb = s.ncol;
        }
        if (!y_rm) { Y = ty; ldy = s.ncol; }
    }
    // up to 32 columns in pairs, rows 16-byte aligned: sixteen lanes per row, sixteen gathers of 16 bytes in flight per lane
    static const bool wide_ok = !(getenv("BDF_SPMM_WIDE") && atoi(getenv("BDF_SPMM_WIDE")) == 0);
    const bool wide = wide_ok && s.ncol >= 2 && s.ncol <= 32 && s.ncol % 2 == 0 && ldb % 2 == 0 && ldy % 2 == 0 && ((uintptr_t)B & 15) == 0 &&
                      ((uintptr_t)Y & 15) == 0 && (!(y_rm && s.Y2) || (((uintptr_t)s.Y2 & 15) == 0));
    if (wide) {
        // a gathered operand of several L2 sizes: one launch per COLUMN PANEL of it (3 MiB: every XCD's L2 holds the panel its workgroups
        // gather from, 23 TB/s of 16-byte lanes instead of the Infinity Cache's 8.6), the rows' running sums carried through Y
        static const bool panels_ok = !(getenv("BDF_SPMM_PANELS") && atoi(getenv("BDF_SPMM_PANELS")) == 0);
        static const int max_panels = getenv("BDF_SPMM_MAX_PANELS") ? atoi(getenv("BDF_SPMM_MAX_PANELS")) : 64;
        const int np = (panels_ok && s.panel_ptr && s.n_panels >= 2 && s.n_panels <= max_panels &&
                        (size_t)s.kin * s.ncol * sizeof(double) >= ((size_t)8 << 20)) ? s.n_panels : 1;
        const dim3 grid((unsigned)((s.m + 15) / 16));
        // ... all panels in ONE launch of a persistent grid (k_spmm_rm16p; BDF_SPMM_FUSED=0: a launch per panel, the rows' running sums
        // carried through Y).  Measured on configuration C5 (profiles/r06_c5_fused_panels.txt, rocprofv3): F p -- 6,250 row blocks,
        // 5 panels -- 5 x 23.7 = 118 us panel by panel, 117 fused; F't -- 3,125 row blocks, 9 panels, every panel launch 2.4
        // generations of workgroups ending on a half-empty chip -- 9 x 15.8 = 142 us against 125-130 fused: 241 us per F'(F p)
        // instead of 260, 1.28 GB of gathered rows per product at 10-11 TB/s (between the Infinity Cache's 8.6 and an L2-resident
        // table's 23: the workgroups drift out of step by a panel or two).
        static const bool fused_ok = !(getenv("BDF_SPMM_FUSED") && atoi(getenv("BDF_SPMM_FUSED")) == 0);
        if (np > 1 && fused_ok) {
            // the panels in one launch: a persistent grid of as many workgroups as the stream's CUs hold (k_spmm_rm16p), every one
            // with KB row blocks' running sums in LDS (4 KB each): the smallest KB whose grid is resident at once
            static int cus = 0;
            if (!cus) {
                hipDeviceProp_t prop;
                BDF_HIP(hipGetDeviceProperties(&prop, ctx->device));
                cus = prop.multiProcessorCount;
            }
            const int avail = ctx->on_reserved ? std::max(1, ctx->reserve_cus) : std::max(1, cus - ctx->reserve_cus);
            const int64_t nblocks = (s.m + 15) / 16;
            const double *bias = y_rm ? s.bias : nullptr;
            double *Y2 = y_rm ? s.Y2 : nullptr;
            for (int64_t rb0 = 0; rb0 < nblocks;) {
                const int64_t left = nblocks - rb0;
                int kb = 1;
                int64_t G = 1;
                for (;; kb++) {
                    int occ = 0;
                    if (s.vals) BDF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmm_rm16p<true>, 256, (size_t)kb * 4096));
                    else BDF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmm_rm16p<false>, 256, (size_t)kb * 4096));
                    G = (int64_t)avail * std::max(1, occ);
                    if (G * kb >= left || kb == 12) break;
                }
                const dim3 pg((unsigned)std::min<int64_t>(G, (left + kb - 1) / kb));
                if (s.vals) hipLaunchKernelGGL(k_spmm_rm16p<true>, pg, dim3(256), (size_t)kb * 4096, ctx->stream, s.m, s.ncol, s.colind, s.vals, B, ldb, Y, ldy,
                                               bias, Y2, ctx->skip_flag, s.panel_ptr, np, kb, rb0, nblocks);
                else hipLaunchKernelGGL(k_spmm_rm16p<false>, pg, dim3(256), (size_t)kb * 4096, ctx->stream, s.m, s.ncol, s.colind, s.vals, B, ldb, Y, ldy,
                                        bias, Y2, ctx->skip_flag, s.panel_ptr, np, kb, rb0, nblocks);
                rb0 += (int64_t)pg.x * kb;
            }
        } else
        for (int p = 0; p < np; p++) {
            const int64_t *pb = np > 1 ? s.panel_ptr + (size_t)p * s.m : nullptr, *pe = np > 1 ? s.panel_ptr + (size_t)(p + 1) * s.m : nullptr;
            const bool last = p == np - 1;
            const double *bias = (last && y_rm) ? s.bias : nullptr;
            double *Y2 = (last && y_rm) ? s.Y2 : nullptr;
            if (s.vals) hipLaunchKernelGGL(k_spmm_rm16<true>, grid, dim3(256), 0, ctx->stream, s.m, s.ncol, s.rowptr, s.colind, s.vals, B, ldb, Y, ldy,
                                           bias, Y2, ctx->skip_flag, pb, pe, p == 0 ? 1 : 0);
            else hipLaunchKernelGGL(k_spmm_rm16<false>, grid, dim3(256), 0, ctx->stream, s.m, s.ncol, s.rowptr, s.colind, s.vals, B, ldb, Y, ldy,
                                    bias, Y2, ctx->skip_flag, pb, pe, p == 0 ? 1 : 0);
        }
    } else
    hipLaunchKernelGGL(k_spmm_rm, dim3((unsigned)((s.m + 7) / 8)), dim3(256), 0, ctx->stream, s.m, s.ncol, s.rowptr, s.colind, s.vals,
                       B, ldb, Y, ldy, y_rm ? s.bias : nullptr, y_rm ? s.Y2 : nullptr, ctx->skip_flag);
    if (!y_rm)
        hipLaunchKernelGGL(k_from_rowmajor, dim3((unsigned)((s.m + 31) / 32), (unsigned)((s.ncol + 31) / 32)), dim3(256), 0,
                           ctx->stream, s.m, s.ncol, (const double *)Y, s.Y, s.yrs, s.ycs, s.bias, s.Y2, ctx->skip_flag);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

// Y = op(F) B for any feature kind.  B(i,c) at B[i*brs + c*bcs], Y(r,c) at Y[r*yrs + c*ycs].
int feat_apply(bdf_ctx *ctx, const bdf_feat *f, bool transpose, const double *B, int64_t brs, int64_t bcs, int ncol,
               double *Y, int64_t yrs, int64_t ycs, const double *bias = nullptr, double *Y2 = nullptr)
{
    if (f->kind == 0 && ncol <= 64 && f->m > 0 && f->n > 0) return dense_apply(ctx, f, transpose, B, brs, bcs, ncol, Y, yrs, ycs, bias, Y2);
    if (f->kind == 0) {
        GemmArgs g;
        g.M = transpose ? f->n : f->m; g.N = ncol; g.K = transpose ? f->m : f->n;
        g.A = f->dense_dev;
        g.ars = transpose ? f->m : 1; g.acs = transpose ? 1 : f->m;
        g.B = B; g.brs = brs; g.bcs = bcs; g.C = Y; g.crs = yrs; g.ccs = ycs; g.bias = bias; g.C2 = Y2;
        return gemm(ctx, g);
    }
    SpmmArgs s;
    s.m = transpose ? f->n : f->m; s.kin = transpose ? f->m : f->n; s.ncol = ncol;
    s.rowptr = transpose ? f->colptr_dev : f->rowptr_dev;
    s.colind = transpose ? f->rowind_dev : f->colind_dev;
    s.vals = f->kind == 1 ? (transpose ? f->cvals_dev : f->rvals_dev) : nullptr;
    s.B = B; s.brs = brs; s.bcs = bcs; s.Y = Y; s.yrs = yrs; s.ycs = ycs; s.bias = bias; s.Y2 = Y2;
    s.panel_ptr = transpose ? f->panel_tr_dev : f->panel_fwd_dev; s.n_panels = transpose ? f->n_panels_tr : f->n_panels_fwd;
    return spmm(ctx, s);
}

// ---- elementwise helpers -------------------------------------------------------------------------------------
__global__ void k_axpy_lambda(int64_t n, double lambda, const double *x, double *y)   // y += lambda x
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fma(lambda, x[i], y[i]);
}

// ---- noise rows e ~ N(0, Lambda^-1) = chol(inv(Lambda))' z  (sampling.jl:298-300) ----------------------------------
// step 1 (one wave): factor the index-reversed Lambda (wave_linalg.h): Lr = [Ah rows, masked | 1/p | sqrt(p)]
template <int DP>
__global__ __launch_bounds__(64) void k_noise_prep(int D, const double *Lambda, double *Lr, int *flag)
{
    __shared__ double tri[WL<DP>::TRI + 64];
    const int lane = threadIdx.x;
    const int c = lane % DP;
    const int ej = D - 1 - c;
    double col[DP];
#pragma unroll
    for (int i = 0; i < DP; i++) {
        const int ei = D - 1 - i;
        double w = (i == c) ? 1.0 : 0.0;
        if (ei >= 0 && ej >= 0) {
            const int lo = ei < ej ? ei : ej, hi = ei < ej ? ej : ei;    // Symmetric(Lambda): upper triangle
            w = Lambda[lo + (int64_t)hi * D];
        }
        col[i] = w;
    }
    double p_own, rp_own;
    if (wl_factor<DP, true>(col, p_own, rp_own, tri, lane) && lane == 0) atomicOr_system(flag, 4);
    if (lane < DP) {
#pragma unroll
        for (int k = 0; k < DP; k++) Lr[c * DP + k] = col[k];
        Lr[DP * DP + c] = rp_own;
        Lr[DP * DP + DP + c] = p_own * fast_rsqrt(p_own);
    }
}

// step 2: T[:,i] = (sample[:,i] - mu) + e_i   (sample == NULL: T[:,i] = scale * e_i);
// e solves U' e = z, i.e. L~' e~ = z~ in reversed coordinates:  e~_j = (sqrt(p_j) z~_j - sum_{m>j} Ah[m][j] e~_m) / p_j
constexpr int NOISE_RPW = 16;       // rows per workgroup of k_noise_rows
template <int DP>
__global__ __launch_bounds__(256) void k_noise_rows(int D, int64_t n, const double *Lr, const double *sample,
                                                     const double *mu, const double *scale_sq, uint64_t seed,
                                                     uint32_t sweep, uint32_t purpose, uint32_t entity, double *T,
                                                     const int32_t *__restrict__ row_ids)
{
    // row_ids (nullable): the row's ORIGINAL id, which keys its noise stream (rows stored at internal positions when several
    // GPUs share the entity); negative = a row nobody owns: no noise (its feature row is zero)
    // NOISE_RPW rows per workgroup.  Phase 1, all 256 threads: the rows' normals (a Philox block + log + sin/cos per pair is ~40x
    // the arithmetic of the solve: one or two pairs per thread, so that 500 rows already fill 32 workgroups), scaled by
    // sqrt(p_j), into LDS.  Phase 2, DP lanes per row: the substitution.
    __shared__ double sL[DP * DP + 2 * DP];
    __shared__ double sz[NOISE_RPW][DP + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < DP * DP + 2 * DP; e += 256) sL[e] = Lr[e];
    const int64_t r0 = (int64_t)blockIdx.x * NOISE_RPW;
    const int npairs = (D + 1) / 2;
    for (int e = tid; e < NOISE_RPW * DP; e += 256) sz[e / DP][e % DP] = 0.0;
    __syncthreads();
    for (int e = tid; e < NOISE_RPW * npairs; e += 256) {
        const int lr = e / npairs, pr = e % npairs;
        const int64_t i = r0 + lr;
        if (i >= n) continue;
        const int64_t rid = row_ids ? (int64_t)row_ids[i] : i;
        if (rid < 0) continue;
        const u32x4 o = bdf_draw(seed, sweep, purpose, entity, (uint64_t)rid, (uint32_t)pr);
        const double u1 = bdf_u01(o.x, o.y), u2 = bdf_u01(o.z, o.w);
        const double r = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925286766559 * u2;
        // normal number ej of the row belongs to reversed position j = D - 1 - ej
        const int e0 = 2 * pr, e1 = 2 * pr + 1;
        sz[lr][D - 1 - e0] = r * cos(t) * sL[DP * DP + DP + (D - 1 - e0)];
        if (e1 < D) sz[lr][D - 1 - e1] = r * sin(t) * sL[DP * DP + DP + (D - 1 - e1)];
    }
    __syncthreads();
    // the substitution, DP lanes per row (lane = reversed position): as soon as e~_j is final every lane m < j takes its term
    // Ah[j][m] e~_j -- one broadcast and one fma per step instead of a dot product walked by a single thread per row
    const int sub = tid % DP, grp = tid / DP;
    const double scale = scale_sq ? sqrt(*scale_sq) : 1.0;
    for (int lr = grp; lr < NOISE_RPW; lr += 256 / DP) {
        const int64_t i = r0 + lr;
        if (i >= n) break;
        double sv = sz[lr][sub];
        for (int j = DP - 1; j >= 0; j--) {
            const double ej = __shfl(sv, j, DP) * sL[DP * DP + j];
            if (sub == j) sv = ej;
            else if (sub < j) sv = fma(-sL[j * DP + sub], ej, sv);
        }
        const int ej = D - 1 - sub;
        if (ej >= 0) {
            const int64_t off = i * D + ej;
            T[off] = sample ? (sample[off] - mu[ej]) + sv : scale * sv;
        }
    }
}

template <int DP>
int noise_rows(bdf_ctx *ctx, int D, int64_t n, const double *Lambda, double *Lr, const double *sample, const double *mu,
               const double *scale_sq, uint32_t purpose, uint32_t entity, double *T, bool prep, const int32_t *row_ids = nullptr)
{
    if (prep) {
        hipLaunchKernelGGL(k_noise_prep<DP>, dim3(1), dim3(64), 0, ctx->stream, D, Lambda, Lr, ctx->flag_dev);
        BDF_HIP(hipGetLastError());
    }
    if (n > 0) {
        hipLaunchKernelGGL(k_noise_rows<DP>, dim3((unsigned)((n + NOISE_RPW - 1) / NOISE_RPW)), dim3(256), 0, ctx->stream, D, n, Lr,
                           sample, mu, scale_sq, ctx->seed, ctx->sweep_host, purpose, entity, T, row_ids);
        BDF_HIP(hipGetLastError());
    }
    return BDF_OK;
}

// rhs(f,d) = FtT(f,d) + E2s(d,f)   (E2s is D x numF: row f of the noise is contiguous)
__global__ void k_add_e2(int64_t numF, int D, const double *E2s, double *rhs)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= numF * D) return;
    const int64_t f = idx % numF;
    const int d = (int)(idx / numF);
    rhs[idx] += E2s[f * D + d];
}

// ---- batched CG --------------------------------------------------------------------------------------------------
struct CgState {
    int64_t n; int D;
    double *X, *R, *P, *Z;               // n x D column-major
    double *bknum, *bkden, *tolb;        // D
    int *active, *iters, *nactive;
    int *done_blocks;                    // columns (workgroups) that have finished the current k_cg_step
    volatile uint64_t *status;           // host-mapped: [0] = generation << 32 | last completed iteration, [1] = active columns
    uint32_t gen;
    int *flag;                           // BDF_WARN_CG_MAXITER: a column still active after the last iteration
};

__device__ __forceinline__ double block_sum(double v, double *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w];
    return s;
}

// The CG kernels take one workgroup per column (no grid-wide reduction); long columns get 1024 threads (CG_THREADS_LONG)
// (bar, nullable: the hand-over counter of k_cg_resident, zeroed here)
__global__ __launch_bounds__(1024) void k_cg_init(CgState s, const double *rhs, double tol, unsigned *bar)
{
    __shared__ double red[16];
    const int d = blockIdx.x;
    const int64_t off = (int64_t)d * s.n;
    double nb = 0.0;
    for (int64_t i = threadIdx.x; i < s.n; i += blockDim.x) {
        const double b = rhs[off + i];
        s.X[off + i] = 0.0; s.R[off + i] = b; s.P[off + i] = b;
        nb = fma(b, b, nb);
    }
    nb = block_sum(nb, red);
    if (threadIdx.x == 0) {
        s.tolb[d] = tol * sqrt(nb);      // tol = tol * norm(b), parallel_cg.jl:65
        s.bkden[d] = 0.0; s.active[d] = 1; s.iters[d] = 0;
        if (d == 0) { *s.nactive = s.D; *s.done_blocks = 0; if (bar) *bar = 0u; }
    }
}

// top of iteration `iter` (1-based): residual check, direction update (parallel_cg.jl:74-83)
__device__ __forceinline__ void cg_pre(const CgState &s, int iter, double *red, int &go)
{
    const int d = blockIdx.x;
    if (!s.active[d]) return;
    const int64_t off = (int64_t)d * s.n;
    double bknum = 0.0;
    for (int64_t i = threadIdx.x; i < s.n; i += blockDim.x) bknum = fma(s.R[off + i], s.R[off + i], bknum);
    bknum = block_sum(bknum, red);
    if (threadIdx.x == 0) {
        go = !(sqrt(bknum) < s.tolb[d]);
        if (!go) { s.active[d] = 0; atomicSub(s.nactive, 1); }
    }
    __syncthreads();
    if (!go) return;
    if (iter > 1) {
        const double bk = bknum / s.bkden[d];
        for (int64_t i = threadIdx.x; i < s.n; i += blockDim.x) s.P[off + i] = fma(bk, s.P[off + i], s.R[off + i]);
    }
    __syncthreads();
    if (threadIdx.x == 0) { s.bkden[d] = bknum; s.bknum[d] = bknum; s.iters[d] = iter; }
}

__global__ __launch_bounds__(1024) void k_cg_pre(CgState s, int iter)
{
    __shared__ double red[16];
    __shared__ int go;
    cg_pre(s, iter, red, go);
}

// bottom of the iteration: z = Z + lambda p; ak = bknum / (z.p); x += ak p; r -= ak z (parallel_cg.jl:85-91)
__device__ __forceinline__ void cg_post(const CgState &s, const double *lambda_p, int iter, double *red)
{
    const int d = blockIdx.x;
    if (!s.active[d] || s.iters[d] != iter) return;
    const double lambda = *lambda_p;
    const int64_t off = (int64_t)d * s.n;
    double zp = 0.0;
    for (int64_t i = threadIdx.x; i < s.n; i += blockDim.x) {
        const double p = s.P[off + i];
        const double z = fma(lambda, p, s.Z[off + i]);
        s.Z[off + i] = z;
        zp = fma(z, p, zp);
    }
    zp = block_sum(zp, red);
    const double ak = s.bknum[d] / zp;
    for (int64_t i = threadIdx.x; i < s.n; i += blockDim.x) {
        s.X[off + i] = fma(ak, s.P[off + i], s.X[off + i]);
        s.R[off + i] = fma(-ak, s.Z[off + i], s.R[off + i]);
    }
}

// bottom of iteration `iter` and top of iteration `iter + 1` in one launch (a column is one workgroup in both)
__global__ __launch_bounds__(1024) void k_cg_step(CgState s, const double *lambda_p, int iter, int maxiter)
{
    __shared__ double red[16];
    __shared__ int go;
    if (*s.nactive == 0) {                // every column has stopped: a launch the host had enqueued ahead
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            s.status[1] = 0;
            __threadfence_system();
            s.status[0] = ((uint64_t)s.gen << 32) | (uint32_t)iter;
        }
        return;
    }
    cg_post(s, lambda_p, iter, red);
    __threadfence_block();
    __syncthreads();
    if (iter < maxiter) cg_pre(s, iter + 1, red, go);
    else if (threadIdx.x == 0 && s.active[blockIdx.x] && s.iters[blockIdx.x] == iter) atomicOr_system(s.flag, (int)BDF_WARN_CG_MAXITER);
    // the last column to finish reports (iteration, active columns) to the host, which enqueues ahead of the device and
    // stops when it reads 0 active columns: no stream synchronisation inside the solve
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(s.done_blocks, 1) == s.D - 1) {
            *s.done_blocks = 0;
            const int na = __hip_atomic_load(s.nactive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s.status[1] = (uint64_t)(iter < maxiter ? na : 0);
            __threadfence_system();
            s.status[0] = ((uint64_t)s.gen << 32) | (uint32_t)iter;
        }
    }
}

// k_cg_step for short columns (n <= 256 EPT): the column's p, z, x, r are read ONCE into registers, both halves of the step
// run on them, and what changed is written once -- the general kernel walks the column four times, each walk a global-memory
// round trip (8.6 us per iteration at n = 500, where the arithmetic is nothing).  Same operations in the same order per element;
// the dot products are summed thread-strided as in block_sum's callers.
template <int EPT>
__global__ __launch_bounds__(256) void k_cg_step_short(CgState s, const double *lambda_p, int iter, int maxiter)
{
    __shared__ double red[16];
    __shared__ int go;
    if (*s.nactive == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            s.status[1] = 0;
            __threadfence_system();
            s.status[0] = ((uint64_t)s.gen << 32) | (uint32_t)iter;
        }
        return;
    }
    const int d = blockIdx.x, tid = threadIdx.x;
    const int64_t off = (int64_t)d * s.n;
    const bool mine = s.active[d] && s.iters[d] == iter;
    if (mine) {
        const double lambda = *lambda_p;
        double p[EPT], z[EPT], x[EPT], r[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int64_t i = tid + 256 * e;
            const bool ok = i < s.n;
            p[e] = ok ? s.P[off + i] : 0.0; z[e] = ok ? s.Z[off + i] : 0.0;
            x[e] = ok ? s.X[off + i] : 0.0; r[e] = ok ? s.R[off + i] : 0.0;
        }
        double zp = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            z[e] = fma(lambda, p[e], z[e]);
            zp = fma(z[e], p[e], zp);
        }
        zp = block_sum(zp, red);
        const double ak = s.bknum[d] / zp;
        double bknum = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            x[e] = fma(ak, p[e], x[e]);
            r[e] = fma(-ak, z[e], r[e]);
            bknum = fma(r[e], r[e], bknum);
        }
        bool proceed = false;
        if (iter >= maxiter && tid == 0) atomicOr_system(s.flag, (int)BDF_WARN_CG_MAXITER);
        if (iter < maxiter) {                              // top of iteration iter + 1 (cg_pre)
            bknum = block_sum(bknum, red);
            if (tid == 0) {
                go = !(sqrt(bknum) < s.tolb[d]);
                if (!go) { s.active[d] = 0; atomicSub(s.nactive, 1); }
            }
            __syncthreads();
            proceed = go != 0;
            if (proceed) {
                const double bk = bknum / s.bkden[d];
#pragma unroll
                for (int e = 0; e < EPT; e++) p[e] = fma(bk, p[e], r[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int64_t i = tid + 256 * e;
            if (i < s.n) {
                s.X[off + i] = x[e]; s.R[off + i] = r[e];
                if (proceed) s.P[off + i] = p[e];
            }
        }
        __syncthreads();
        if (proceed && tid == 0) { s.bkden[d] = bknum; s.bknum[d] = bknum; s.iters[d] = iter + 1; }
    }
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(s.done_blocks, 1) == s.D - 1) {
            *s.done_blocks = 0;
            const int na = __hip_atomic_load(s.nactive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s.status[1] = (uint64_t)(iter < maxiter ? na : 0);
            __threadfence_system();
            s.status[0] = ((uint64_t)s.gen << 32) | (uint32_t)iter;
        }
    }
}

// ---- the whole solve in ONE launch for a small resident operator (F'F of at most 512 features, at most 32 columns) -----------------
// An iteration of the batched solve is two dependent launches (product 10.4 us, step 7.2 us at numF = 500, D = 32), and both
// are the floor of a dependent launch of a few workgroups (~5 us) plus a little work: fifteen iterations are 0.26 of configuration
// C3's 0.54 ms.  Here ceil(numF / 16) workgroups stay resident for the whole solve.  Workgroup w is (a) the owner of the rows
// 16 w .. 16 w + 15 of the operator -- its waves keep their quarter of K of those rows in REGISTERS as matrix operands across all
// iterations -- and (b) the owner of column w of the solve: that column's p, x, r and scalars live in its registers.  An
// iteration: every workgroup multiplies its rows into all columns of P (read from memory past the caches) and writes its rows of Z
// write-through; a grid-wide hand-over; the column owners run EXACTLY k_cg_step_short's arithmetic on their column (same sums in
// the same order) and write the new p write-through; a second hand-over.  The hand-overs are a monotonic counter (arrive after
// the wave's write-through stores have completed, poll with agent-scope loads); nothing is fenced: what crosses workgroups is
// written with write-through stores and read with agent-scope loads.  All workgroups are co-resident (at most 32 of 256 threads).
struct CgResident {
    CgState s;
    const double *FF;                 // n x n, column-major, symmetric
    const double *lambda_p;
    int maxiter, nwg;
    unsigned *bar;                    // zeroed before the launch
};

#ifdef BDF_CG_STAMPS      // diagnostic build (tools/c3_cg_stamps.py): workgroup 0's clock (s_memrealtime, 100 MHz) at eight points of every iteration
__device__ unsigned long long g_cgstamps[64 * 8];
#define CGSTAMP(it, k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (it) < 64) g_cgstamps[(it) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CGSTAMP(it, k) do { } while (0)
#endif

__device__ __forceinline__ void cg_grid_sync(unsigned *bar, unsigned target, int *flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through stores have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 21)) { atomicOr_system(flag, 16); break; }       // bounded (~0.2 s): a workgroup that is not resident must not hang the device
        }
    }
    __syncthreads();
}

template <int CB>
__global__ __launch_bounds__(256) void k_cg_resident(CgResident c)
{
    __shared__ double red[3][CB][4][64];
    __shared__ double sred[16];
    __shared__ int go;
    // P staged for the product: column c at Pl + c * PSTR (517: an odd stride, the sixteen columns of a matrix operand in sixteen banks)
    constexpr int PSTR = 517, PLD = 32 * CB;               // n <= 512 rows, 16 CB columns: at most 32 CB elements per thread
    __shared__ double Pl[16 * CB * PSTR + 2];
    const CgState &s = c.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, h = lane >> 4;
    const int w = blockIdx.x;
    const int64_t n = s.n;
    const int D = s.D;
    // (a) this workgroup's rows of the operator: every wave keeps its quarter of K of rows 16 w + i as matrix operands
    // (k_dense_nn's split of K over the waves and its assignment of k to lanes and matrix instructions: the same sums in the same
    // order, so the iterates -- and the iteration counts -- are those of the two-launch solve to the last bit)
    constexpr int KS = 32;
    const int64_t kq = ((n + 3) / 4 + 15) / 16 * 16;               // <= 128 = 4 KS for n <= 512
    const int64_t row = (int64_t)w * 16 + i, kb = (int64_t)wave * kq, ke = (kb + kq < n) ? kb + kq : n;
    double a[KS];
#pragma unroll
    for (int t = 0; t < KS; t++) {
        const int64_t k = kb + 16 * (t >> 2) + 4 * h + (t & 3);
        a[t] = (row < n && k < ke) ? c.FF[k + row * n] : 0.0;      // (symmetric: row `row` is the contiguous column `row`)
    }
    // (b) column w of the solve (k_cg_init and k_cg_pre(1) have run: x = 0, r = p = b, bknum = bkden = |b|^2, iters = 1)
    const int d = w;
    const bool owner = d < D;
    const int64_t off = (int64_t)d * n;
    constexpr int EPT = 2;
    double p[EPT], x[EPT], r[EPT];
    bool active = false;
    double bknum_d = 0.0, bkden_d = 0.0, tolb_d = 0.0;
    int iters_d = 0;
    if (owner) {
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int64_t q = tid + 256 * e;
            const bool ok = q < n;
            p[e] = ok ? s.P[off + q] : 0.0; x[e] = ok ? s.X[off + q] : 0.0; r[e] = ok ? s.R[off + q] : 0.0;
        }
        active = s.active[d] != 0; bknum_d = s.bknum[d]; bkden_d = s.bkden[d]; tolb_d = s.tolb[d]; iters_d = s.iters[d];
    }
    const double lambda = *c.lambda_p;
    unsigned sync_no = 0;
    for (int iter = 1; iter <= c.maxiter; iter++) {
        if (__hip_atomic_load(s.nactive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) break;      // (the same value in every workgroup: read after a hand-over)
        CGSTAMP(iter, 0);
        // ---- Z[rows of w, :] = FF[rows of w, :] P
        // P -- n x D, the columns one after the other: 128 KB at n = 500, D = 32, written by the other workgroups a moment ago -- comes
        // into LDS by COALESCED loads past the L2, all of them in flight at once (a thread's elements are 256 apart), and the matrix
        // operands are read from there.  (Until round 6 every lane fetched its operands itself, 8 bytes at a stride of a column:
        // sixty-four cache lines per instruction -- 7.25 us of an iteration's 12.9, profiles/r06_c3_cg_handover.txt.)  Same operand
        // values into the same matrix instructions in the same order: the iterates are unchanged to the last bit.
        {
            const int64_t total = n * (int64_t)D;
            double pv[PLD];
            // (no branches: an element beyond the end reads the last one again and is not stored)
#pragma unroll
            for (int q = 0; q < PLD; q++) {
                const int64_t e = tid + 256 * q;
                pv[q] = __hip_atomic_load(s.P + (e < total ? e : total - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int col = 0;
            int k = tid;
#pragma unroll
            for (int q = 0; q < PLD; q++) {
                // (n >= 128 -- cg_solve takes this kernel for no smaller operator -- : at most two columns' ends per 256 elements)
                const bool w1 = k >= (int)n;
                k -= w1 ? (int)n : 0; col += w1 ? 1 : 0;
                const bool w2 = k >= (int)n;
                k -= w2 ? (int)n : 0; col += w2 ? 1 : 0;
                Pl[col < D ? col * PSTR + k : 16 * CB * PSTR] = pv[q];               // (beyond the end: a spare slot)
                k += 256;
            }
        }
        __syncthreads();
        fd4 acc[CB];
#pragma unroll
        for (int cb = 0; cb < CB; cb++) acc[cb] = fd4{0.0, 0.0, 0.0, 0.0};
        {
#pragma unroll
            for (int t = 0; t < KS; t++) {
                const int64_t k = kb + 16 * (t >> 2) + 4 * h + (t & 3);
#pragma unroll
                for (int cb = 0; cb < CB; cb++) {
                    const int col = 16 * cb + i;
                    const double b = (k < ke && col < D) ? Pl[col * PSTR + k] : 0.0;
                    acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b, acc[cb], 0, 0, 0);
                }
            }
        }
        CGSTAMP(iter, 1);                  // P loaded (128 KB past the L2, agent scope), the matrix instructions issued
        if (wave > 0) {
#pragma unroll
            for (int cb = 0; cb < CB; cb++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) red[wave - 1][cb][rr][lane] = acc[cb][rr];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int cb = 0; cb < CB; cb++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const double v = ((acc[cb][rr] + red[0][cb][rr][lane]) + red[1][cb][rr][lane]) + red[2][cb][rr][lane];
                    const int64_t zr = (int64_t)w * 16 + h + 4 * rr;
                    const int col = 16 * cb + i;
                    if (zr < n && col < D) __hip_atomic_store(s.Z + zr + (int64_t)col * n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        CGSTAMP(iter, 2);                  // the waves' sums added, Z's rows stored (write-through, not yet drained)
        cg_grid_sync(c.bar, ++sync_no * (unsigned)c.nwg, s.flag);
        CGSTAMP(iter, 3);                  // first hand-over passed: every workgroup's rows of Z are in memory
        // ---- the step of column w: bottom of iteration `iter`, top of iteration `iter + 1` (k_cg_step_short's arithmetic)
        if (owner && active && iters_d == iter) {           // (workgroup-uniform)
            double z[EPT];
            double zp = 0.0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int64_t q = tid + 256 * e;
                z[e] = q < n ? __hip_atomic_load(s.Z + off + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                z[e] = fma(lambda, p[e], z[e]);
                zp = fma(z[e], p[e], zp);
            }
            zp = block_sum(zp, sred);
            const double ak = bknum_d / zp;
            double bknum = 0.0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                x[e] = fma(ak, p[e], x[e]);
                r[e] = fma(-ak, z[e], r[e]);
                bknum = fma(r[e], r[e], bknum);
            }
            if (iter >= c.maxiter) {
                if (tid == 0) atomicOr_system(s.flag, (int)BDF_WARN_CG_MAXITER);
            } else {
                bknum = block_sum(bknum, sred);
                if (tid == 0) go = !(sqrt(bknum) < tolb_d);
                __syncthreads();
                if (go) {
                    const double bk = bknum / bkden_d;
#pragma unroll
                    for (int e = 0; e < EPT; e++) {
                        p[e] = fma(bk, p[e], r[e]);
                        const int64_t q = tid + 256 * e;
                        if (q < n) __hip_atomic_store(s.P + off + q, p[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    bkden_d = bknum; bknum_d = bknum; iters_d = iter + 1;
                } else {
                    active = false;
                    if (tid == 0) __hip_atomic_fetch_sub(s.nactive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();                          // (`go` is rewritten in the next iteration)
            }
        }
        CGSTAMP(iter, 4);                  // column w's step: Z's column read, two block sums, p stored
        cg_grid_sync(c.bar, ++sync_no * (unsigned)c.nwg, s.flag);
        CGSTAMP(iter, 5);                  // second hand-over passed: every column's new p is in memory
    }
    if (owner) {
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int64_t q = tid + 256 * e;
            if (q < n) { s.X[off + q] = x[e]; s.R[off + q] = r[e]; }
        }
        if (tid == 0) { s.active[d] = active ? 1 : 0; s.iters[d] = iters_d; s.bknum[d] = bknum_d; s.bkden[d] = bkden_d; }
    }
}

#ifdef BDF_CG_STAMPS
extern "C" int bdf_debug_cg_stamps(unsigned long long *host512)
{
    BDF_HIP(hipDeviceSynchronize());
    BDF_HIP(hipMemcpyFromSymbol(host512, HIP_SYMBOL(g_cgstamps), sizeof(unsigned long long) * 64 * 8));
    return BDF_OK;
}
#endif

// k_cg_step for long columns (n > 2048): one workgroup per column is one CU's bandwidth per column (82 us per iteration at
// n = 50,000, D = 32: 32 CUs moving 100 MB).  Here a column is cut into G chunks, grid (D, G), and the step becomes three
// launches with the two dot products summed over the chunks in chunk order by every workgroup that needs them:
//   a: z = Z + lambda p, partial z.p          b: ak; x += ak p; r -= ak z; partial r.r          c: stop test; p = bk p + r
// bkden is double-buffered by iteration parity (slot 1 = s.bkden, written by k_cg_pre at iteration 1; slot 0 = bkden0): in c
// every workgroup of a column reads the old value while chunk 0 writes the new one.
struct CgChunks { int G; int64_t len; double *partA, *partB, *bkden0; };

__device__ __forceinline__ bool cg_all_stopped(const CgState &s, int iter)
{
    if (*s.nactive != 0) return false;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        s.status[1] = 0;
        __threadfence_system();
        s.status[0] = ((uint64_t)s.gen << 32) | (uint32_t)iter;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_cg_long_a(CgState s, CgChunks c, const double *lambda_p, int iter)
{
    __shared__ double red[16];
    if (cg_all_stopped(s, iter)) return;
    const int d = blockIdx.x, g = blockIdx.y;
    if (!s.active[d] || s.iters[d] != iter) return;
    const double lambda = *lambda_p;
    const int64_t off = (int64_t)d * s.n, i0 = g * c.len, i1 = (i0 + c.len < s.n) ? i0 + c.len : s.n;
    double zp = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const double p = s.P[off + i];
        const double z = fma(lambda, p, s.Z[off + i]);
        s.Z[off + i] = z;
        zp = fma(z, p, zp);
    }
    zp = block_sum(zp, red);
    if (threadIdx.x == 0) c.partA[d * c.G + g] = zp;
}

__global__ __launch_bounds__(256) void k_cg_long_b(CgState s, CgChunks c, int iter)
{
    __shared__ double red[16];
    if (*s.nactive == 0) return;
    const int d = blockIdx.x, g = blockIdx.y;
    if (!s.active[d] || s.iters[d] != iter) return;
    double zp = 0.0;
    for (int q = 0; q < c.G; q++) zp += c.partA[d * c.G + q];
    const double ak = s.bknum[d] / zp;
    const int64_t off = (int64_t)d * s.n, i0 = g * c.len, i1 = (i0 + c.len < s.n) ? i0 + c.len : s.n;
    double rr = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        s.X[off + i] = fma(ak, s.P[off + i], s.X[off + i]);
        const double r = fma(-ak, s.Z[off + i], s.R[off + i]);
        s.R[off + i] = r;
        rr = fma(r, r, rr);
    }
    rr = block_sum(rr, red);
    if (threadIdx.x == 0) c.partB[d * c.G + g] = rr;
}

__global__ __launch_bounds__(256) void k_cg_long_c(CgState s, CgChunks c, int iter, int maxiter)
{
    if (*s.nactive == 0) return;                          // (a) has reported
    const int d = blockIdx.x, g = blockIdx.y;
    if (s.active[d] && s.iters[d] == iter && iter >= maxiter && g == 0 && threadIdx.x == 0) atomicOr_system(s.flag, (int)BDF_WARN_CG_MAXITER);
    if (s.active[d] && s.iters[d] == iter && iter < maxiter) {      // top of iteration iter + 1 (cg_pre)
        double rr = 0.0;
        for (int q = 0; q < c.G; q++) rr += c.partB[d * c.G + q];
        const bool go = !(sqrt(rr) < s.tolb[d]);
        double *bk_old = (iter & 1) ? s.bkden : c.bkden0, *bk_new = (iter & 1) ? c.bkden0 : s.bkden;
        if (go) {
            const double bk = rr / bk_old[d];
            const int64_t off = (int64_t)d * s.n, i0 = g * c.len, i1 = (i0 + c.len < s.n) ? i0 + c.len : s.n;
            for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) s.P[off + i] = fma(bk, s.P[off + i], s.R[off + i]);
        }
        __syncthreads();                                  // every thread has read active / iters
        if (g == 0 && threadIdx.x == 0) {
            if (!go) { s.active[d] = 0; atomicSub(s.nactive, 1); }
            else { bk_new[d] = rr; s.bknum[d] = rr; s.iters[d] = iter + 1; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(s.done_blocks, 1) == s.D * c.G - 1) {
            *s.done_blocks = 0;
            const int na = __hip_atomic_load(s.nactive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s.status[1] = (uint64_t)(iter < maxiter ? na : 0);
            __threadfence_system();
            s.status[0] = ((uint64_t)s.gen << 32) | (uint32_t)iter;
        }
    }
}

// ---- the same solve with its state ROW-MAJOR (element (i, d) at [i * D + d]) -- sparse features (round 6).  The sparse products gather
// ROWS of their dense operand (256 contiguous bytes at D = 32), so with the state column-major, as the reference's matrices are,
// every F'(F p) was wrapped in two tiled transposes (k_to_rowmajor / k_from_rowmajor: 912 + 912 launches per sweep of configuration
// C5).  Here P, Z, R, X live row-major from the solve's first launch to its last: the products take and leave them as they are, and
// the three vector steps take a chunk of rows per workgroup, a thread per (row, column) with the column fastest: 32 lanes read one
// row.  A column's dot products are summed over the chunk's rows in row order by the eight row lanes of a column, then over the
// chunks in chunk order (another order than the column-major kernels': the iterates agree to rounding, not to the bit).
//   rm_init: R = P = b (row-major copy made by k_to_rowmajor), X = 0, partial |b|^2        rm_start: tol |b|, bknum = bkden = |b|^2, iters = 1
//   rm_a / rm_b / rm_c: k_cg_long_a / _b / _c's arithmetic
struct CgRm { int G; int64_t len; double *partA, *partB, *bkden0, *zp, *rrs; };
#define BDF_CG_RM_MAXG 1024

__device__ __forceinline__ double rm_colsum(double v, double (*red)[32], int d, int rl)
{
    // the eight row lanes of column d, added in row-lane order (every thread gets the sum)
    __syncthreads();
    red[rl][d] = v;
    __syncthreads();
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) sum += red[q][d];
    return sum;
}

// this workgroup's per-column partial to part[d * G + g]; the workgroup that finishes LAST adds the G partials of every column --
// row lane rl those of chunks rl, rl + 8, ..., then the eight row lanes in order: a fixed order whichever workgroup it is -- and
// returns true in it (with the column's sum in `total`)
__device__ __forceinline__ bool rm_reduce(const CgState &s, const CgRm &c, double *part, double v, bool keep, double (*red)[32], int d, int rl, double &total)
{
    __shared__ int last;
    const double mine = rm_colsum(v, red, d, rl);
    if (rl == 0 && keep) __hip_atomic_store(part + d * c.G + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(s.done_blocks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.G - 1;
    __syncthreads();
    if (!last) return false;
    if (threadIdx.x == 0) __hip_atomic_store(s.done_blocks, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the partials were written through by the other workgroups: one acquire, then plain loads -- many in flight; taken one by one
    // past the L2 the ~50 loads of a row lane were ~50 round trips: 55 us per iteration)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double acc = 0.0;
    if (keep) {
        const double *pp = part + d * c.G;
#pragma unroll 8
        for (int q = rl; q < c.G; q += 8) acc += pp[q];
    }
    total = rm_colsum(acc, red, d, rl);
    return true;
}

// the G partials of every column added by THIS workgroup (row lane rl those of chunks rl, rl + 8, ..., then the row lanes in order: the
// same fixed order in every workgroup) -- the vector steps read their dot products this way: a kernel boundary lies between the
// partials' writers and their readers, nothing to wait for
__device__ __forceinline__ double rm_sum_parts(const CgRm &c, const double *part, bool keep, double (*red)[32], int d, int rl)
{
    double acc = 0.0;
    if (keep) {
        const double *pp = part + d * c.G;
#pragma unroll 8
        for (int q = rl; q < c.G; q += 8) acc += pp[q];
    }
    return rm_colsum(acc, red, d, rl);
}

__global__ __launch_bounds__(256) void k_cg_rm_init(CgState s, CgRm c, double tol)
{
    __shared__ double red[8][32];
    const int d = threadIdx.x & 31, rl = threadIdx.x >> 5, g = blockIdx.x;
    const bool dok = d < s.D;
    const int64_t i0 = g * c.len, i1 = (i0 + c.len < s.n) ? i0 + c.len : s.n;
    double nb = 0.0;
    if (dok)
        for (int64_t i = i0 + rl; i < i1; i += 8) {
            const double b = s.R[i * s.D + d];
            s.P[i * s.D + d] = b; s.X[i * s.D + d] = 0.0;
            nb = fma(b, b, nb);
        }
    double tot = 0.0;
    if (!rm_reduce(s, c, c.partA, nb, dok, red, d, rl, tot)) return;
    if (rl == 0 && dok) {
        s.tolb[d] = tol * sqrt(tot);                     // tol = tol * norm(b), parallel_cg.jl:65
        const bool go = !(sqrt(tot) < s.tolb[d]);        // top of iteration 1 (cg_pre): the residual is b
        s.active[d] = go ? 1 : 0; s.iters[d] = go ? 1 : 0;
        s.bkden[d] = tot; s.bknum[d] = tot;
        if (!go) atomicSub(s.nactive, 1);
    }
}

__global__ __launch_bounds__(256) void k_cg_rm_a(CgState s, CgRm c, const double *lambda_p, int iter)
{
    __shared__ double red[8][32];
    if (cg_all_stopped(s, iter)) return;
    const int d = threadIdx.x & 31, rl = threadIdx.x >> 5, g = blockIdx.x;
    const bool act = d < s.D && s.active[d] && s.iters[d] == iter;
    const double lambda = *lambda_p;
    const int64_t i0 = g * c.len, i1 = (i0 + c.len < s.n) ? i0 + c.len : s.n;
    double zp = 0.0;
    if (act)
#pragma unroll 4
        for (int64_t i = i0 + rl; i < i1; i += 8) {
            const double p = s.P[i * s.D + d];
            const double z = fma(lambda, p, s.Z[i * s.D + d]);
            s.Z[i * s.D + d] = z;
            zp = fma(z, p, zp);
        }
    zp = rm_colsum(zp, red, d, rl);
    if (rl == 0 && act) c.partA[d * c.G + g] = zp;
}

__global__ __launch_bounds__(256) void k_cg_rm_b(CgState s, CgRm c, int iter)
{
    __shared__ double red[8][32];
    if (*s.nactive == 0) return;
    const int d = threadIdx.x & 31, rl = threadIdx.x >> 5, g = blockIdx.x;
    const bool act = d < s.D && s.active[d] && s.iters[d] == iter;
    double rr = 0.0;
    const double zp = rm_sum_parts(c, c.partA, act, red, d, rl);
    if (act) {
        const double ak = s.bknum[d] / zp;
        const int64_t i0 = g * c.len, i1 = (i0 + c.len < s.n) ? i0 + c.len : s.n;
#pragma unroll 4
        for (int64_t i = i0 + rl; i < i1; i += 8) {
            const int64_t e = i * s.D + d;
            s.X[e] = fma(ak, s.P[e], s.X[e]);
            const double r = fma(-ak, s.Z[e], s.R[e]);
            s.R[e] = r;
            rr = fma(r, r, rr);
        }
    }
    rr = rm_colsum(rr, red, d, rl);
    if (rl == 0 && act) c.partB[d * c.G + g] = rr;
}

__global__ __launch_bounds__(256) void k_cg_rm_c(CgState s, CgRm c, int iter, int maxiter)
{
    if (*s.nactive == 0) return;                          // (a) has reported
    const int d = threadIdx.x & 31, rl = threadIdx.x >> 5, g = blockIdx.x;
    const bool act = d < s.D && s.active[d] && s.iters[d] == iter;
    if (act && iter >= maxiter && g == 0 && rl == 0) atomicOr_system(s.flag, (int)BDF_WARN_CG_MAXITER);
    bool go = false;
    __shared__ double red[8][32];
    const double rr = rm_sum_parts(c, c.partB, act && iter < maxiter, red, d, rl);
    if (act && iter < maxiter) {                          // top of iteration iter + 1 (cg_pre)
        go = !(sqrt(rr) < s.tolb[d]);
        if (go) {
            const double bk = rr / s.bkden[d];
            const int64_t i0 = g * c.len, i1 = (i0 + c.len < s.n) ? i0 + c.len : s.n;
    #pragma unroll 4
        for (int64_t i = i0 + rl; i < i1; i += 8) { const int64_t e = i * s.D + d; s.P[e] = fma(bk, s.P[e], s.R[e]); }
        }
    }
    // the columns' bookkeeping by the workgroup that FINISHES LAST (every other one has read active / iters / bkden by then)
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(s.done_blocks, 1) == c.G - 1;
    }
    __syncthreads();
    if (!last) return;
    if (rl == 0 && act && iter < maxiter) {
        if (!go) { s.active[d] = 0; atomicSub(s.nactive, 1); }
        else { s.bkden[d] = rr; s.bknum[d] = rr; s.iters[d] = iter + 1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *s.done_blocks = 0;
        __threadfence();
        const int na = __hip_atomic_load(s.nactive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s.status[1] = (uint64_t)(iter < maxiter ? na : 0);
        __threadfence_system();
        s.status[0] = ((uint64_t)s.gen << 32) | (uint32_t)iter;
    }
}

__global__ void k_cg_rm_zero(CgState s)
{
    if (threadIdx.x == 0) { *s.nactive = s.D; *s.done_blocks = 0; }
}

// ---- beta' beta, trace(beta'beta Lambda), lambda_beta ~ Gamma ----------------------------------------------------
// G = beta' beta (D x D) by one block
__global__ __launch_bounds__(256) void k_btb(int D, int64_t numF, const double *beta, double *G)
{
    for (int e = threadIdx.x; e < D * D; e += blockDim.x) {
        const int i = e % D, j = e / D;
        double s = 0.0;
        for (int64_t f = 0; f < numF; f++) s = fma(beta[f + (int64_t)i * numF], beta[f + (int64_t)j * numF], s);
        G[e] = s;
    }
}

__global__ void k_tinv_feat(int D, const double *G, const double *WI, const double *lambda_beta, double *Tinv)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < D * D) Tinv[e] = WI[e] + G[e] * (*lambda_beta);      // Tinv += beta'beta * lambda_beta, macau.jl:128
}

__global__ __launch_bounds__(64) void k_lambda_beta(int D, int64_t numF, const double *G, const double *Lambda, double nu,
                                                    double mu, uint64_t seed, uint32_t sweep, uint32_t entity,
                                                    double *lambda_beta)
{
    // trace((beta'beta) Lambda) = sum_ij G[i][j] Lambda[j][i]
    double tr = 0.0;
    for (int e = threadIdx.x; e < D * D; e += 64) {
        const int i = e % D, j = e / D;
        tr = fma(G[i + j * D], Lambda[j + i * D], tr);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tr += __shfl_xor(tr, off);
    if (threadIdx.x == 0) {
        const double nux = nu + (double)numF * (double)D;
        const double mux = mu * nux / (nu + mu * tr);
        *lambda_beta = bdf_gamma(seed, sweep, entity, (uint64_t)D, 0.5 * nux) * (2.0 * mux / nux);
    }
}

// ---- direct solve for numF <= 64: (FF + lambda I) X = RHS on one wave (solve_full, sampling.jl:314-320) -------------
template <int DP>
__global__ __launch_bounds__(64) void k_solve_small(int n, int ncol, const double *FF, const double *lambda_p,
                                                    const double *rhs, double *X, int *flag)
{
    __shared__ double tri[WL<DP>::TRI + 64];
    const int lane = threadIdx.x;
    const int c = lane % DP;
    const double lambda = *lambda_p;
    double rowm[DP];
#pragma unroll
    for (int i = 0; i < DP; i++) {
        double w = (i == c) ? 1.0 : 0.0;
        if (i < n && c < n) w = FF[i + (int64_t)c * n] + ((i == c) ? lambda : 0.0);
        rowm[i] = w;
    }
    double p_own, rp_own;
    if (wl_factor<DP, true>(rowm, p_own, rp_own, tri, lane) && lane == 0) atomicOr_system(flag, 8);
    for (int q = 0; q < ncol; q++) {
        double b = (lane < DP && c < n) ? rhs[c + (int64_t)q * n] : 0.0;
        b = wl_forward<DP>(rowm, b, rp_own, lane);       // b' = wh p;  yh = w sqrt(p) = b'
        b = wl_backward<DP, true>(tri, b, rp_own, lane);
        if (lane < DP && c < n) X[c + (int64_t)q * n] = b;
    }
}

int ensure_dense(bdf_feat *f)
{
    if (f->dense_dev) return BDF_OK;
    bdf_ctx *ctx = f->ctx;
    BDF_REQUIRE((double)f->m * (double)f->n * 8.0 <= 4e9, BDF_ERR_ARG,
                "FF path needs F'F of a sparse F with %lld x %lld entries: too large, use the CG path (compute_ff_size)",
                (long long)f->m, (long long)f->n);
    // densify by applying F to the identity: dense(:, j) = F e_j
    size_t nn = (size_t)f->n * (size_t)f->n;
    double *eye;
    BDF_HIP(hipMalloc((void **)&eye, std::max<size_t>(nn * sizeof(double), 8)));
    std::vector<double> h(nn, 0.0);
    for (int64_t j = 0; j < f->n; j++) h[(size_t)j * f->n + j] = 1.0;
    BDF_HIP(hipMemcpy(eye, h.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    double *dense;
    BDF_HIP(hipMalloc((void **)&dense, std::max<size_t>((size_t)f->m * f->n * sizeof(double), 8)));
    int rc = feat_apply(ctx, f, false, eye, 1, f->n, (int)f->n, dense, 1, f->m);
    if (rc) return rc;
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    BDF_HIP(hipFree(eye));
    f->dense_dev = dense;
    return BDF_OK;
}

int ensure_FF(bdf_feat *f)
{
    if (f->FF_dev) return BDF_OK;
    bdf_ctx *ctx = f->ctx;
    int rc = ensure_dense(f);
    if (rc) return rc;
    BDF_HIP(hipMalloc((void **)&f->FF_dev, std::max<size_t>((size_t)f->n * f->n * sizeof(double), 8)));
    GemmArgs g;                          // FF = full(At_mul_B(F, F)), RelationData.jl:338
    g.M = f->n; g.N = f->n; g.K = f->m;
    g.A = f->dense_dev; g.ars = f->m; g.acs = 1;
    g.B = f->dense_dev; g.brs = 1; g.bcs = f->m;
    g.C = f->FF_dev; g.crs = 1; g.ccs = f->n; g.bias = nullptr; g.C2 = nullptr;
    return gemm(ctx, g);
}

template <typename T>
int upload_vec(const std::vector<T> &v, T **dptr)
{
    BDF_HIP(hipMalloc((void **)dptr, std::max<size_t>(v.size() * sizeof(T), 8)));
    if (!v.empty()) BDF_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return BDF_OK;
}

int create_sparse(bdf_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int32_t *rows, const int32_t *cols,
                  const double *vals, bdf_feat **out)
{
    BDF_REQUIRE(ctx && out, BDF_ERR_ARG, "bdf_feat_create: NULL argument");
    BDF_REQUIRE(m >= 0 && n >= 0 && nnz >= 0 && (nnz == 0 || (rows && cols)), BDF_ERR_ARG, "bdf_feat_create: bad argument");
    for (int64_t q = 0; q < nnz; q++) {
        BDF_REQUIRE(rows[q] >= 1 && rows[q] <= m && cols[q] >= 1 && cols[q] <= n, BDF_ERR_BOUNDS,
                    "bdf_feat_create: entry %lld (%d,%d) outside %lld x %lld", (long long)q, rows[q], cols[q], (long long)m, (long long)n);
    }
    BDF_HIP(hipSetDevice(ctx->device));
    auto build = [&](const int32_t *major, const int32_t *minor, int64_t nmajor, std::vector<int64_t> &ptr,
                     std::vector<int32_t> &ind, std::vector<double> &v) {
        ptr.assign((size_t)nmajor + 1, 0);
        for (int64_t q = 0; q < nnz; q++) ptr[(size_t)major[q]]++;
        for (int64_t j = 0; j < nmajor; j++) ptr[(size_t)j + 1] += ptr[(size_t)j];
        std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
        ind.assign((size_t)nnz, 0);
        if (vals) v.assign((size_t)nnz, 0.0);
        for (int64_t q = 0; q < nnz; q++) {          // stable in input order (sortperm, sparsebin_csr.jl:23)
            int64_t dst = cur[(size_t)major[q] - 1]++;
            ind[(size_t)dst] = minor[q] - 1;
            if (vals) v[(size_t)dst] = vals[q];
        }
    };
    bdf_feat *f = new bdf_feat();
    memset(f, 0, sizeof(*f));
    struct Guard { bdf_feat *f; ~Guard() { if (f) bdf_feat_destroy(f); } } guard{f};        // error paths free what was uploaded
    f->ctx = ctx; f->kind = vals ? 1 : 2; f->m = m; f->n = n; f->nnz = nnz;
    std::vector<int64_t> ptr; std::vector<int32_t> ind; std::vector<double> v;
    int rc;
    // column panels (spmm): where each row's entries cross a multiple of BDF_SPMM_PANEL_ROWS columns -- only when every row's entries
    // are in column order (the panels must keep the order of the row's sum) and the operand is large enough to be worth it
    auto panels = [&](int64_t nmajor, int64_t nminor, int64_t **dev, int *np_out) -> int {
        const int64_t P = (nminor + BDF_SPMM_PANEL_ROWS - 1) / BDF_SPMM_PANEL_ROWS;
        *dev = nullptr; *np_out = 0;
        if (P < 2 || P > 64 || (size_t)(P + 1) * nmajor * sizeof(int64_t) > ((size_t)256 << 20)) return BDF_OK;
        std::vector<int64_t> pp((size_t)(P + 1) * nmajor);
        for (int64_t r = 0; r < nmajor; r++) {
            int64_t q = ptr[(size_t)r];
            const int64_t e = ptr[(size_t)r + 1];
            for (int64_t k = q + 1; k < e; k++)
                if (ind[(size_t)k] < ind[(size_t)k - 1]) return BDF_OK;           // not in column order: one pass
            for (int64_t p = 0; p <= P; p++) {
                while (q < e && ind[(size_t)q] < p * BDF_SPMM_PANEL_ROWS) q++;
                pp[(size_t)p * nmajor + r] = (p == P) ? e : q;
            }
        }
        int rc2 = upload_vec(pp, dev);
        if (!rc2) *np_out = (int)P;
        return rc2;
    };
    build(rows, cols, m, ptr, ind, v);
    if ((rc = upload_vec(ptr, &f->rowptr_dev)) || (rc = upload_vec(ind, &f->colind_dev))) return rc;
    if (vals && (rc = upload_vec(v, &f->rvals_dev))) return rc;
    if ((rc = panels(m, n, &f->panel_fwd_dev, &f->n_panels_fwd))) return rc;
    build(cols, rows, n, ptr, ind, v);
    if ((rc = upload_vec(ptr, &f->colptr_dev)) || (rc = upload_vec(ind, &f->rowind_dev))) return rc;
    if (vals && (rc = upload_vec(v, &f->cvals_dev))) return rc;
    if ((rc = panels(n, m, &f->panel_tr_dev, &f->n_panels_tr))) return rc;
    guard.f = nullptr;
    *out = f;
    return BDF_OK;
}

}  // namespace

extern "C" int bdf_feat_create_dense(bdf_ctx *ctx, int64_t m, int64_t n, const double *F, bdf_feat **out)
{
    BDF_REQUIRE(ctx && out && m >= 0 && n >= 0 && (m * n == 0 || F), BDF_ERR_ARG, "bdf_feat_create_dense: bad argument");
    BDF_HIP(hipSetDevice(ctx->device));
    bdf_feat *f = new bdf_feat();
    memset(f, 0, sizeof(*f));
    f->ctx = ctx; f->kind = 0; f->m = m; f->n = n; f->nnz = m * n;
    struct Guard { bdf_feat *f; ~Guard() { if (f) bdf_feat_destroy(f); } } guard{f};
    BDF_HIP(hipMalloc((void **)&f->dense_dev, std::max<size_t>((size_t)m * n * sizeof(double), 8)));
    if (m * n) BDF_HIP(hipMemcpy(f->dense_dev, F, (size_t)m * n * sizeof(double), hipMemcpyHostToDevice));
    guard.f = nullptr;
    *out = f;
    return BDF_OK;
}

extern "C" int bdf_feat_create_csr(bdf_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int32_t *rows,
                                   const int32_t *cols, const double *vals, bdf_feat **out)
{
    BDF_REQUIRE(nnz == 0 || vals, BDF_ERR_ARG, "bdf_feat_create_csr: vals is NULL");
    static const double one = 1.0;
    return create_sparse(ctx, m, n, nnz, rows, cols, nnz ? vals : &one, out);
}

extern "C" int bdf_feat_create_bin(bdf_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int32_t *rows,
                                   const int32_t *cols, bdf_feat **out)
{
    return create_sparse(ctx, m, n, nnz, rows, cols, nullptr, out);
}

extern "C" int bdf_feat_destroy(bdf_feat *f)
{
    if (!f) return BDF_OK;
    hipSetDevice(f->ctx->device);
    hipStreamSynchronize(f->ctx->stream);
    hipFree(f->dense_dev); hipFree(f->rowptr_dev); hipFree(f->colind_dev); hipFree(f->rvals_dev);
    hipFree(f->colptr_dev); hipFree(f->rowind_dev); hipFree(f->cvals_dev); hipFree(f->FF_dev); hipFree(f->chol_ws);
    hipFree(f->panel_fwd_dev); hipFree(f->panel_tr_dev);
    hipFree(f->row_ids_dev); hipFree(f->gather_dev);
    if (f->eig_Q) { hipFree(f->eig_Q->dense_dev); delete f->eig_Q; }
    hipFree(f->eig_s); hipFree(f->eig_y);
    delete f;
    return BDF_OK;
}

extern "C" int bdf_feat_set_row_ids(bdf_feat *f, const int32_t *row_ids_host)
{
    BDF_REQUIRE(f, BDF_ERR_ARG, "bdf_feat_set_row_ids: NULL argument");
    if (f->row_ids_dev) { BDF_HIP(hipFree(f->row_ids_dev)); f->row_ids_dev = nullptr; }
    if (row_ids_host && f->m > 0) {
        BDF_HIP(hipMalloc((void **)&f->row_ids_dev, (size_t)f->m * sizeof(int32_t)));
        BDF_HIP(hipMemcpy(f->row_ids_dev, row_ids_host, (size_t)f->m * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return BDF_OK;
}

extern "C" int bdf_feat_size(const bdf_feat *f, int64_t *m, int64_t *n, int64_t *nnz)
{
    BDF_REQUIRE(f && m && n && nnz, BDF_ERR_ARG, "bdf_feat_size: NULL argument");
    *m = f->m; *n = f->n; *nnz = f->nnz;
    return BDF_OK;
}

extern "C" int bdf_feat_mul(bdf_ctx *ctx, const bdf_feat *f, const double *B, int ncol, double *out, int transpose)
{
    BDF_REQUIRE(ctx && f && B && out && ncol >= 1, BDF_ERR_ARG, "bdf_feat_mul: bad argument");
    const int64_t kin = transpose ? f->m : f->n, kout = transpose ? f->n : f->m;
    return feat_apply(ctx, f, transpose != 0, B, 1, kin, ncol, out, 1, kout);
}

extern "C" int bdf_feat_AtA_mul(bdf_ctx *ctx, const bdf_feat *f, const double *X, int ncol, double lambda, double *out)
{
    BDF_REQUIRE(ctx && f && X && out && ncol >= 1, BDF_ERR_ARG, "bdf_feat_AtA_mul: bad argument");
    void *tmp;
    int rc = bdf_scratch(ctx, (size_t)f->m * ncol * sizeof(double), &tmp);
    if (rc) return rc;
    if ((rc = feat_apply(ctx, f, false, X, 1, f->n, ncol, (double *)tmp, 1, f->m))) return rc;
    if ((rc = feat_apply(ctx, f, true, (const double *)tmp, 1, f->m, ncol, out, 1, f->n))) return rc;
    const int64_t tot = f->n * ncol;
    hipLaunchKernelGGL(k_axpy_lambda, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tot, lambda, X, out);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

extern "C" int bdf_uhat(bdf_ctx *ctx, const bdf_feat *f, int D, const double *beta, const double *mu,
                        double *uhat_out, double *mu_matrix_out)
{
    BDF_REQUIRE(ctx && f && beta && uhat_out, BDF_ERR_ARG, "bdf_uhat: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_uhat: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    BDF_REQUIRE(!mu_matrix_out || mu, BDF_ERR_ARG, "bdf_uhat: mu is NULL");
    // (F beta)(i,d) written at uhat[d + i*D]
    return feat_apply(ctx, f, false, beta, 1, f->n, D, uhat_out, D, 1, mu, mu_matrix_out);
}

extern "C" int bdf_hyper_feature_terms(bdf_ctx *ctx, int D, int64_t numF, const double *beta, const double *WI,
                                       const double *lambda_beta_dev, double *Tinv_out)
{
    BDF_REQUIRE(ctx && beta && WI && lambda_beta_dev && Tinv_out, BDF_ERR_ARG, "bdf_hyper_feature_terms: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_feature_terms: bad num_latent");
    void *G;
    int rc = bdf_scratch(ctx, (size_t)D * D * sizeof(double), &G);
    if (rc) return rc;
    {   // G = beta' beta (D x D) through the split-K product
        GemmArgs g;
        g.M = D; g.N = D; g.K = numF; g.A = beta; g.ars = numF; g.acs = 1; g.B = beta; g.brs = 1; g.bcs = numF;
        g.C = (double *)G; g.crs = 1; g.ccs = D; g.bias = nullptr; g.C2 = nullptr;
        int rcg = gemm(ctx, g);
        if (rcg) return rcg;
    }
    hipLaunchKernelGGL(k_tinv_feat, dim3((D * D + 255) / 256), dim3(256), 0, ctx->stream, D, (const double *)G, WI,
                       lambda_beta_dev, Tinv_out);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

// D simultaneous cg_AtA solves of (F'F + lambda I) X = rhs (solve_cg2, parallel_matrix.jl:488-507); with use_ff the operator
// is the precomputed F'F.  R, P, Z: numF x D; Tm: N x D; scal: 3 D doubles; ints: 2 D + 1 ints.
// ctx->cg_part: the partial dot products of the chunked steps -- the column-major kernels' (2 x BDF_MAX_D x 64 + BDF_MAX_D) or the
// row-major ones' (2 x 32 x BDF_CG_RM_MAXG + 128)
constexpr size_t CG_PART_DOUBLES = (size_t)2 * 32 * BDF_CG_RM_MAXG + 128 > (size_t)2 * BDF_MAX_D * 64 + BDF_MAX_D ? (size_t)2 * 32 * BDF_CG_RM_MAXG + 128 : (size_t)2 * BDF_MAX_D * 64 + BDF_MAX_D;
static int cg_solve(bdf_ctx *ctx, bdf_feat *f, bool use_ff, int D, const double *lambda_beta_dev, const double *rhs,
                    double *beta_out, double tol, int maxiter, double *R, double *P, double *Z, double *Tm, double *scal,
                    int *ints, int **iters_dev, double *Xrm = nullptr /* numF x D spare (the row-major solve's X), or NULL */)
{
    const int64_t N = f->m, numF = f->n;
    int rc;
    CgState s;
    s.n = numF; s.D = D; s.X = beta_out; s.R = R; s.P = P; s.Z = Z;
    s.bknum = scal; s.bkden = scal + D; s.tolb = scal + 2 * D;
    s.active = ints; s.iters = ints + D; s.nactive = ints + 2 * D;
    s.done_blocks = ints + 2 * D + 1;
    if (!ctx->cg_status) {
        BDF_HIP(hipHostMalloc((void **)&ctx->cg_status, 2 * sizeof(uint64_t), hipHostMallocMapped));
        ctx->cg_status[0] = ctx->cg_status[1] = 0;
    }
    struct SkipGuard { bdf_ctx *c; ~SkipGuard() { c->skip_flag = nullptr; } } guard{ctx};
    s.status = ctx->cg_status;
    s.flag = ctx->flag_dev;
    s.gen = ++ctx->cg_gen;
    const dim3 cgb(numF >= 8192 ? 1024 : 256);      // threads per column
    // a small resident operator: the whole solve in one launch (k_cg_resident; BDF_CG_RESIDENT=0: the two launches per iteration)
    static const bool resident_ok = !(getenv("BDF_CG_RESIDENT") && atoi(getenv("BDF_CG_RESIDENT")) == 0);
    bool resident = resident_ok && use_ff && numF >= 128 && numF <= 512 && D <= 32 && D <= (numF + 15) / 16;
    if (resident) {
        // its workgroups hand over through a counter they all poll: ALL ceil(numF / 16) of them must be resident at once.  One
        // workgroup per CU is what the kernel's registers and LDS allow for certain (asked of the runtime below), so the stream
        // needs that many CUs: not the reserved hyperprior stream (a handful of CUs), and the row context's CUs minus the
        // reserved ones.  (A caller-supplied CU-masked stream the library cannot see is covered by the spin bound: flag 16.)
        static int cus = 0, occ32 = -1, occ16 = -1;
        if (!cus) {
            hipDeviceProp_t prop;
            BDF_HIP(hipGetDeviceProperties(&prop, ctx->device));
            cus = prop.multiProcessorCount;
            BDF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ16, k_cg_resident<1>, 256, 0));
            BDF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ32, k_cg_resident<2>, 256, 0));
        }
        const int avail = ctx->on_reserved ? 0 : std::max(0, cus - ctx->reserve_cus);
        const int per_cu = std::min(1, D <= 16 ? occ16 : occ32);
        resident = (int64_t)per_cu * avail >= (numF + 15) / 16;
    }
    if (resident && !ctx->cg_bar) BDF_HIP(hipMalloc((void **)&ctx->cg_bar, sizeof(unsigned)));
    // sparse features, long columns: the state ROW-MAJOR from the first launch to the last (k_cg_rm_*): no transposes around the products
    static const bool rm_ok = !(getenv("BDF_CG_ROWMAJOR") && atoi(getenv("BDF_CG_ROWMAJOR")) == 0);
    static const bool long_ok_ = !(getenv("BDF_CG_LONG") && atoi(getenv("BDF_CG_LONG")) == 0);
    const bool rm = rm_ok && long_ok_ && Xrm && !use_ff && f->kind != 0 && numF > 2048 && D <= 32;
    CgRm cr;
    cr.len = 256;                                             // rows per workgroup: thirty-two per row lane (G ~ 200 at 50,000 rows: every workgroup adds G partials per column)
    if (const char *ev = getenv("BDF_CG_RM_LEN")) { const int v = atoi(ev); if (v >= 8 && v % 8 == 0) cr.len = v; }      // A/B hook (the dots' order follows it)
    cr.G = (int)((numF + cr.len - 1) / cr.len);
    if (cr.G > BDF_CG_RM_MAXG) { cr.G = BDF_CG_RM_MAXG; cr.len = (numF + cr.G - 1) / cr.G; cr.G = (int)((numF + cr.len - 1) / cr.len); }
    cr.partA = cr.partB = cr.bkden0 = cr.zp = cr.rrs = nullptr;
    if (rm) {
        if (!ctx->cg_part) BDF_HIP(hipMalloc((void **)&ctx->cg_part, CG_PART_DOUBLES * sizeof(double)));
        cr.partA = ctx->cg_part; cr.partB = cr.partA + (size_t)32 * BDF_CG_RM_MAXG; cr.zp = cr.partB + (size_t)32 * BDF_CG_RM_MAXG; cr.rrs = cr.zp + 64;
        // (X row-major in the caller's spare buffer: beta_out is column-major and receives the solution at the end)
        hipLaunchKernelGGL(k_cg_rm_zero, dim3(1), dim3(64), 0, ctx->stream, s);
        hipLaunchKernelGGL(k_to_rowmajor, dim3((unsigned)((numF + 31) / 32), (unsigned)((D + 31) / 32)), dim3(256), 0, ctx->stream, numF, D, rhs, (int64_t)1, numF, R,
                           (const int *)nullptr);
        s.X = Xrm;
        hipLaunchKernelGGL(k_cg_rm_init, dim3(cr.G), dim3(256), 0, ctx->stream, s, cr, tol);
        BDF_HIP(hipGetLastError());
    } else {
    hipLaunchKernelGGL(k_cg_init, dim3(D), cgb, 0, ctx->stream, s, (const double *)rhs, tol, resident ? ctx->cg_bar : (unsigned *)nullptr);
    BDF_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_cg_pre, dim3(D), cgb, 0, ctx->stream, s, 1);
    }
    // The host enqueues iterations AHEAD of the device (no stream synchronisation: the device never idles between
    // iterations) and reads the (iteration, active columns) word the device writes to host-mapped memory after every
    // iteration.  Run-ahead is bounded to CG_AHEAD iterations; once every column has stopped, the launches already enqueued
    // return at once (product kernels through ctx->skip_flag, k_cg_step by itself).
    // long columns: G chunks per column (k_cg_long_*); the partial dot products live behind the spare scalars
    static const bool long_ok = !(getenv("BDF_CG_LONG") && atoi(getenv("BDF_CG_LONG")) == 0);
    const bool long_cols = long_ok && numF > 2048;
    CgChunks ch;
    ch.G = (int)std::min<int64_t>(64, (numF + 4095) / 4096);
    ch.len = (numF + ch.G - 1) / ch.G;
    ch.partA = ch.partB = ch.bkden0 = nullptr;
    if (long_cols) {
        if (!ctx->cg_part) BDF_HIP(hipMalloc((void **)&ctx->cg_part, CG_PART_DOUBLES * sizeof(double)));     // (the scratch buffers are reused by the products inside the loop)
        ch.partA = ctx->cg_part; ch.partB = ch.partA + (size_t)BDF_MAX_D * 64; ch.bkden0 = ch.partB + (size_t)BDF_MAX_D * 64;
    }
    if (resident) {
        CgResident c;
        c.s = s; c.FF = f->FF_dev; c.lambda_p = lambda_beta_dev; c.maxiter = maxiter; c.nwg = (int)((numF + 15) / 16); c.bar = ctx->cg_bar;
        if (D <= 16) hipLaunchKernelGGL(k_cg_resident<1>, dim3(c.nwg), dim3(256), 0, ctx->stream, c);
        else hipLaunchKernelGGL(k_cg_resident<2>, dim3(c.nwg), dim3(256), 0, ctx->stream, c);
        BDF_HIP(hipGetLastError());
        *iters_dev = s.iters;
        return BDF_OK;
    }
    constexpr int CG_AHEAD = 3;
    ctx->skip_flag = s.nactive;
    for (int iter = 1; iter <= maxiter; iter++) {
        if (iter > CG_AHEAD) {
            const uint64_t want = ((uint64_t)s.gen << 32) | (uint32_t)(iter - CG_AHEAD);
            uint64_t st;
            long spins = 0;
            while ((st = s.status[0]) < want || (st >> 32) != s.gen) {
                if (++spins > 2000000000L || ((spins & 0xfffff) == 0 && hipStreamQuery(ctx->stream) != hipErrorNotReady)) {
                    // the stream drained without the report (a device fault): stop enqueuing
                    st = s.status[0];
                    if (st < want || (st >> 32) != s.gen) { bdf_set_error("cg_solve: the device did not report iteration %d", iter - CG_AHEAD); return BDF_ERR_HIP; }
                    break;
                }
            }
            if (s.status[1] == 0) break;
        }
        if (use_ff && D <= 64) {
            if ((rc = dense_nn(ctx, f->FF_dev, numF, numF, P, 1, numF, D, Z, 1, numF, nullptr, nullptr))) return rc;
        } else if (use_ff) {
            GemmArgs g;
            g.M = numF; g.N = D; g.K = numF; g.A = f->FF_dev; g.ars = 1; g.acs = numF;
            g.B = P; g.brs = 1; g.bcs = numF; g.C = Z; g.crs = 1; g.ccs = numF; g.bias = nullptr; g.C2 = nullptr;
            if ((rc = gemm(ctx, g))) return rc;
        } else {
            // the N x D intermediate row-major: contiguous writes of the first product, contiguous operand rows of the second
            if (rm) {
                if ((rc = feat_apply(ctx, f, false, P, D, 1, D, Tm, D, 1))) return rc;
                if ((rc = feat_apply(ctx, f, true, Tm, D, 1, D, Z, D, 1))) return rc;
            } else {
                if ((rc = feat_apply(ctx, f, false, P, 1, numF, D, Tm, D, 1))) return rc;
                if ((rc = feat_apply(ctx, f, true, Tm, D, 1, D, Z, 1, numF))) return rc;
            }
        }
        // bottom of this iteration and top of the next in one launch
        if (numF <= 512) hipLaunchKernelGGL(k_cg_step_short<2>, dim3(D), dim3(256), 0, ctx->stream, s, (const double *)lambda_beta_dev, iter, maxiter);
        else if (numF <= 1024) hipLaunchKernelGGL(k_cg_step_short<4>, dim3(D), dim3(256), 0, ctx->stream, s, (const double *)lambda_beta_dev, iter, maxiter);
        else if (numF <= 2048) hipLaunchKernelGGL(k_cg_step_short<8>, dim3(D), dim3(256), 0, ctx->stream, s, (const double *)lambda_beta_dev, iter, maxiter);
        else if (rm) {
            hipLaunchKernelGGL(k_cg_rm_a, dim3(cr.G), dim3(256), 0, ctx->stream, s, cr, (const double *)lambda_beta_dev, iter);
            hipLaunchKernelGGL(k_cg_rm_b, dim3(cr.G), dim3(256), 0, ctx->stream, s, cr, iter);
            hipLaunchKernelGGL(k_cg_rm_c, dim3(cr.G), dim3(256), 0, ctx->stream, s, cr, iter, maxiter);
        } else if (long_cols) {
            hipLaunchKernelGGL(k_cg_long_a, dim3(D, ch.G), dim3(256), 0, ctx->stream, s, ch, (const double *)lambda_beta_dev, iter);
            hipLaunchKernelGGL(k_cg_long_b, dim3(D, ch.G), dim3(256), 0, ctx->stream, s, ch, iter);
            hipLaunchKernelGGL(k_cg_long_c, dim3(D, ch.G), dim3(256), 0, ctx->stream, s, ch, iter, maxiter);
        } else hipLaunchKernelGGL(k_cg_step, dim3(D), cgb, 0, ctx->stream, s, (const double *)lambda_beta_dev, iter, maxiter);
        BDF_HIP(hipGetLastError());
    }
    ctx->skip_flag = nullptr;
    if (rm) {
        // the solution, row-major in s.X, into the caller's column-major beta_out
        hipLaunchKernelGGL(k_from_rowmajor, dim3((unsigned)((numF + 31) / 32), (unsigned)((D + 31) / 32)), dim3(256), 0, ctx->stream, numF, D,
                           (const double *)s.X, beta_out, (int64_t)1, numF, (const double *)nullptr, (double *)nullptr, (const int *)nullptr);
        BDF_HIP(hipGetLastError());
    }
    *iters_dev = s.iters;
    return BDF_OK;
}


// ---- direct solve through the eigendecomposition of F'F (solve_full, src/sampling.jl:314-320, for 64 < numF <= BDF_EIG_MAX) ----
// The reference factors FF + lambda I anew in every iteration because lambda_beta is resampled.  F'F itself never changes:
// F'F = Q diag(s) Q' (symmetric, s >= 0) is computed ONCE -- on the host, at first use: Householder tridiagonalisation and
// the implicit QL iteration (the EISPACK tred2 / tql2 procedures) -- and every iteration's solve is two small dense products
// on the matrix cores with a scaling between them: beta = Q ((Q' rhs) ./ (s + lambda)).  Same solution as the factorisation
// to rounding (residual ~1e-13 ||rhs|| on C3's matrices); 0.02 ms instead of 0.34 ms per iteration at numF = 500.
namespace {

// symmetric A (n x n, column-major, both triangles) -> eigenvalues d (ascending), eigenvectors in the columns of V
// returns false if the QL iteration did not converge for some eigenvalue within 200 sweeps (the caller then takes the factorisation)
bool eig_sym_host(int n, const double *A, std::vector<double> &d, std::vector<double> &V)
{
    bool converged = true;
    V.assign(A, A + (size_t)n * n);
    d.assign((size_t)n, 0.0);
    std::vector<double> e((size_t)n, 0.0);
    auto v = [&](int i, int j) -> double & { return V[(size_t)i + (size_t)j * n]; };
    // -- Householder reduction to tridiagonal form (tred2)
    for (int j = 0; j < n; j++) d[j] = v(n - 1, j);
    for (int i = n - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; k++) scale += fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; j++) { d[j] = v(i - 1, j); v(i, j) = 0.0; v(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1], g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; j++) e[j] = 0.0;
            for (int j = 0; j < i; j++) {
                f = d[j];
                v(j, i) = f;
                g = e[j] + v(j, j) * f;
                for (int k = j + 1; k <= i - 1; k++) { g += v(k, j) * d[k]; e[k] += v(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; j++) e[j] -= hh * d[j];
            for (int j = 0; j < i; j++) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; k++) v(k, j) -= (f * e[k] + g * d[k]);
                d[j] = v(i - 1, j);
                v(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; i++) {
        v(n - 1, i) = v(i, i);
        v(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; k++) d[k] = v(k, i + 1) / h;
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
                for (int k = 0; k <= i; k++) g += v(k, i + 1) * v(k, j);
                for (int k = 0; k <= i; k++) v(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; k++) v(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; j++) { d[j] = v(n - 1, j); v(n - 1, j) = 0.0; }
    v(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
    // -- implicit QL iteration on the tridiagonal matrix, accumulating the rotations (tql2)
    for (int i = 1; i < n; i++) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; l++) {
        tst1 = std::max(tst1, fabs(d[l]) + fabs(e[l]));
        int m = l;
        while (m < n) { if (fabs(e[m]) <= eps * tst1) break; m++; }
        if (m > l) {
            int iter = 0;
            do {
                iter++;
                double g = d[l], p = (d[l + 1] - g) / (2.0 * e[l]), r = hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; i++) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; i--) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; k++) {
                        h = v(k, i + 1);
                        v(k, i + 1) = s * v(k, i) + c * h;
                        v(k, i) = c * v(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (fabs(e[l]) > eps * tst1 && iter < 200);
            if (fabs(e[l]) > eps * tst1) converged = false;
        }
        d[l] += f;
        e[l] = 0.0;
    }
    return converged;
}

__global__ void k_eig_scale(int64_t n, int D, const double *s, const double *lambda_p, double *Y, int *flag)      // Y(i, c) /= s_i + lambda
{
    const double lambda = *lambda_p;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * D) {
        const double den = s[t % n] + lambda;          // (s >= 0: clamped when the decomposition was made)
        if (!(den > 0.0) && t < n) atomicOr_system(flag, 8);      // F'F + lambda I not positive definite (what the Cholesky path reports); the flag is host memory
        Y[t] = Y[t] / den;
    }
}

int ensure_eig(bdf_feat *f, int D)
{
    bdf_ctx *ctx = f->ctx;
    const int64_t n = f->n;
    int rc;
    if (!f->eig_Q) {
        if ((rc = ensure_FF(f))) return rc;
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<double> A((size_t)n * n), s, Q;
        BDF_HIP(hipMemcpy(A.data(), f->FF_dev, A.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < n; j++)          // exactly symmetric input (the product's two triangles agree to rounding only)
            for (int64_t i = j + 1; i < n; i++) A[(size_t)i + (size_t)j * n] = A[(size_t)j + (size_t)i * n];
        if (!eig_sym_host((int)n, A.data(), s, Q)) { f->eig_failed = true; return BDF_OK; }      // (the caller falls back to bdf_chol_solve)
        for (double &x : s) x = std::max(x, 0.0);       // F'F is positive semi-definite: an eigenvalue below zero is rounding
        bdf_feat *q = new bdf_feat();
        q->ctx = ctx; q->kind = 0; q->m = n; q->n = n; q->nnz = n * n;
        if (hipMalloc((void **)&q->dense_dev, Q.size() * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&f->eig_s, (size_t)n * sizeof(double)) != hipSuccess) {
            hipFree(q->dense_dev); delete q; hipFree(f->eig_s); f->eig_s = nullptr;
            bdf_set_error("ensure_eig: out of device memory");
            return BDF_ERR_HIP;
        }
        BDF_HIP(hipMemcpy(q->dense_dev, Q.data(), Q.size() * sizeof(double), hipMemcpyHostToDevice));
        BDF_HIP(hipMemcpy(f->eig_s, s.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        f->eig_Q = q;
    }
    if (f->eig_y_cols < D) {
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        if (f->eig_y) BDF_HIP(hipFree(f->eig_y));
        f->eig_y = nullptr; f->eig_y_cols = 0;
        BDF_HIP(hipMalloc((void **)&f->eig_y, (size_t)n * D * sizeof(double)));
        f->eig_y_cols = D;
    }
    return BDF_OK;
}

// beta (n x D column-major) = (F'F + lambda I) \ rhs
int eig_solve(bdf_ctx *ctx, bdf_feat *f, int D, const double *lambda_dev, const double *rhs, double *beta_out)
{
    const int64_t n = f->n;
    int rc;
    if (f->eig_failed) return bdf_chol_solve(ctx, f, D, lambda_dev, rhs, beta_out);
    if ((rc = ensure_eig(f, D))) return rc;
    if (f->eig_failed) return bdf_chol_solve(ctx, f, D, lambda_dev, rhs, beta_out);        // the QL iteration did not converge: factor instead
    if ((rc = feat_apply(ctx, f->eig_Q, true, rhs, 1, n, D, f->eig_y, 1, n))) return rc;              // Y = Q' rhs
    hipLaunchKernelGGL(k_eig_scale, dim3((unsigned)((n * D + 255) / 256)), dim3(256), 0, ctx->stream, n, D, (const double *)f->eig_s, lambda_dev, f->eig_y,
                       ctx->flag_dev);
    BDF_HIP(hipGetLastError());
    return feat_apply(ctx, f->eig_Q, false, f->eig_y, 1, n, D, beta_out, 1, n);                       // beta = Q Y
}

}  // namespace

extern "C" int bdf_sample_beta(bdf_ctx *ctx, const bdf_feat *fc, int D, const double *sample, const double *mu,
                               const double *Lambda, double *lambda_beta_dev, int use_ff, double tol, int maxiter,
                               int sample_lambda, double lb_nu, double lb_mu, uint32_t entity_tag,
                               double *beta_out, double *rhs_out, int32_t *iters_out)
{
    return bdf_sample_beta_ranks(ctx, nullptr, fc, D, sample, mu, Lambda, lambda_beta_dev, use_ff, tol, maxiter, sample_lambda, lb_nu,
                                 lb_mu, entity_tag, beta_out, rhs_out, iters_out);
}

// Several ranks (comm != NULL, conjugate gradients): the D solves are shared out as solve_cg2 shares them over its workers
// (src/parallel_matrix.jl:488-507): every rank forms the whole right-hand side (the same noise streams on every rank: one
// product with F'), solves a contiguous block of ceil(D / P) columns and the blocks are all-gathered.  A column's iterates do
// not depend on which other columns are solved beside it, so beta is the one a single rank computes.  The direct solve
// handles all D right-hand sides in one factorisation and stays whole.
extern "C" int bdf_sample_beta_ranks(bdf_ctx *ctx, bdf_comm *comm, const bdf_feat *fc, int D, const double *sample, const double *mu,
                                     const double *Lambda, double *lambda_beta_dev, int use_ff, double tol, int maxiter,
                                     int sample_lambda, double lb_nu, double lb_mu, uint32_t entity_tag,
                                     double *beta_out, double *rhs_out, int32_t *iters_out)
{
    BDF_REQUIRE(ctx && fc && sample && mu && Lambda && lambda_beta_dev && beta_out, BDF_ERR_ARG, "bdf_sample_beta: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_sample_beta: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    bdf_feat *f = const_cast<bdf_feat *>(fc);
    const int64_t N = f->m, numF = f->n;
    if (std::isnan(tol)) tol = 2.220446049250313e-16 * (double)numF;       // eps()*numF, sampling.jl:294-296
    if (maxiter <= 0) maxiter = (int)numF;
    const int DP = D <= 16 ? 16 : (D <= 32 ? 32 : 64);

    // scratch layout (doubles): Lr | T (D x N) | E2s (D x numF) | rhs | R P Z Tm | scalars
    size_t nLr = (size_t)DP * DP + 2 * DP, nT = (size_t)D * N, nE2 = (size_t)D * numF, nB = (size_t)numF * D, nTm = (size_t)N * D;
    size_t total = nLr + nT + nE2 + nB * 4 + nTm + 3 * (size_t)D + 64 + (size_t)D * D;
    void *sv;
    int rc = bdf_scratch(ctx, total * sizeof(double) + (2 * (size_t)D + 16) * sizeof(int), &sv);      // ints: active D | iters D | nactive | done
    if (rc) return rc;
    double *Lr = (double *)sv, *T = Lr + nLr, *E2s = T + nT, *rhs = E2s + nE2, *R = rhs + nB, *P = R + nB, *Z = P + nB,
           *Tm = Z + nB, *scal = Tm + nTm, *G = scal + 3 * D + 64;
    int *ints = (int *)(G + (size_t)D * D);

    // rhs = F'((sample - mu)' + E1) + sqrt(lb) E2
#define NOISE(DPV)                                                                                                      \
    do {                                                                                                                \
        if ((rc = noise_rows<DPV>(ctx, D, N, Lambda, Lr, sample, mu, nullptr, BDF_P_BETA_E1, entity_tag, T, true, f->row_ids_dev))) return rc; \
        if ((rc = noise_rows<DPV>(ctx, D, numF, Lambda, Lr, nullptr, nullptr, lambda_beta_dev, BDF_P_BETA_E2, entity_tag, E2s, false))) return rc; \
    } while (0)
    if (DP == 16) NOISE(16); else if (DP == 32) NOISE(32); else NOISE(64);
#undef NOISE
    // T holds (target)' as D x N: element (i,d) at T[i*D + d]
    if ((rc = feat_apply(ctx, f, true, T, D, 1, D, rhs, 1, numF))) return rc;
    if (numF * D > 0) {
        hipLaunchKernelGGL(k_add_e2, dim3((unsigned)((numF * D + 255) / 256)), dim3(256), 0, ctx->stream, numF, D, E2s, rhs);
        BDF_HIP(hipGetLastError());
    }
    if (rhs_out) BDF_HIP(hipMemcpyAsync(rhs_out, rhs, nB * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));

    if (use_ff && (rc = ensure_FF(f))) return rc;
    if (use_ff) {
        // solve_full (sampling.jl:314-320): a direct solve, Cholesky on one wave up to 64 features, blocked on the matrix cores above
        if (numF <= 16) hipLaunchKernelGGL(k_solve_small<16>, dim3(1), dim3(64), 0, ctx->stream, (int)numF, D, f->FF_dev, lambda_beta_dev, rhs, beta_out, ctx->flag_dev);
        else if (numF <= 32) hipLaunchKernelGGL(k_solve_small<32>, dim3(1), dim3(64), 0, ctx->stream, (int)numF, D, f->FF_dev, lambda_beta_dev, rhs, beta_out, ctx->flag_dev);
        else if (numF <= 64) hipLaunchKernelGGL(k_solve_small<64>, dim3(1), dim3(64), 0, ctx->stream, (int)numF, D, f->FF_dev, lambda_beta_dev, rhs, beta_out, ctx->flag_dev);
        else if (numF <= BDF_EIG_MAX && !getenv("BDF_NO_EIG")) { if ((rc = eig_solve(ctx, f, D, lambda_beta_dev, rhs, beta_out))) return rc; }
        else if ((rc = bdf_chol_solve(ctx, f, D, lambda_beta_dev, rhs, beta_out))) return rc;
        BDF_HIP(hipGetLastError());
        if (iters_out) BDF_HIP(hipMemsetAsync(iters_out, 0, D * sizeof(int32_t), ctx->stream));
    } else {
        // D simultaneous cg_AtA solves (solve_cg2, parallel_matrix.jl:488-507).  The operator p -> F'(F p) is applied as
        // (F'F) p when F'F is small and cheaper than the two products (numF <= 1024 and numF^2 <= nnz(F): C3's 6040 x 500
        // dense F: 2 MB read per iteration instead of 2 x 24 MB) -- the same operator, formed once per feature matrix
        static const bool cg_ff = !(getenv("BDF_CG_FF") && atoi(getenv("BDF_CG_FF")) == 0);
        const bool ff_op = cg_ff && numF > 0 && numF <= 1024 && numF * numF <= f->nnz;
        if (ff_op && (rc = ensure_FF(f))) return rc;
        int *cg_iters = nullptr;
        int rank = 0, world = 1;
        if (comm && (rc = bdf_comm_size(comm, &rank, &world))) return rc;
        if (world <= 1) {
            if ((rc = cg_solve(ctx, f, ff_op, D, lambda_beta_dev, rhs, beta_out, tol, maxiter, R, P, Z, Tm, scal, ints, &cg_iters, E2s))) return rc;      // (E2s: free once rhs is formed)
            if (iters_out) BDF_HIP(hipMemcpyAsync(iters_out, cg_iters, D * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            // this rank's block of columns, solved into its place of a gather buffer of world blocks (>= D columns), then the
            // exchange; the iteration counts travel behind each block's columns
            const int nc = (D + world - 1) / world, c0 = rank * nc, mine = std::max(0, std::min(nc, D - c0));
            const size_t blk = (size_t)nc * numF * sizeof(double) + (size_t)nc * sizeof(int32_t);
            if (f->gather_bytes < blk * (size_t)world) {
                BDF_HIP(hipStreamSynchronize(ctx->stream));
                if (f->gather_dev) BDF_HIP(hipFree(f->gather_dev));
                f->gather_dev = nullptr; f->gather_bytes = 0;
                BDF_HIP(hipMalloc((void **)&f->gather_dev, blk * (size_t)world));
                f->gather_bytes = blk * (size_t)world;
            }
            char *gb = (char *)f->gather_dev;
            double *my_beta = (double *)(gb + (size_t)rank * blk);
            int32_t *my_iters = (int32_t *)(gb + (size_t)rank * blk + (size_t)nc * numF * sizeof(double));
            BDF_HIP(hipMemsetAsync(my_beta, 0, blk, ctx->stream));
            if (mine > 0) {
                if ((rc = cg_solve(ctx, f, ff_op, mine, lambda_beta_dev, rhs + (size_t)c0 * numF, my_beta, tol, maxiter, R, P, Z, Tm, scal, ints, &cg_iters, E2s))) return rc;
                BDF_HIP(hipMemcpyAsync(my_iters, cg_iters, (size_t)mine * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
            }
            if ((rc = bdf_allgather_block(ctx, comm, gb, blk)) || (rc = bdf_allgather_join(ctx, comm))) return rc;
            for (int r = 0; r < world; r++) {
                const int rc0 = r * nc, rn = std::max(0, std::min(nc, D - rc0));
                if (rn <= 0) break;
                BDF_HIP(hipMemcpyAsync(beta_out + (size_t)rc0 * numF, gb + (size_t)r * blk, (size_t)rn * numF * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                if (iters_out)
                    BDF_HIP(hipMemcpyAsync(iters_out + rc0, gb + (size_t)r * blk + (size_t)nc * numF * sizeof(double), (size_t)rn * sizeof(int32_t),
                                           hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
    }
    if (sample_lambda) {
        {
            GemmArgs g;
            g.M = D; g.N = D; g.K = numF; g.A = beta_out; g.ars = numF; g.acs = 1; g.B = beta_out; g.brs = 1; g.bcs = numF;
            g.C = G; g.crs = 1; g.ccs = D; g.bias = nullptr; g.C2 = nullptr;
            if ((rc = gemm(ctx, g))) return rc;
        }
        hipLaunchKernelGGL(k_lambda_beta, dim3(1), dim3(64), 0, ctx->stream, D, numF, (const double *)G, Lambda, lb_nu, lb_mu,
                           ctx->seed, ctx->sweep_host, entity_tag, lambda_beta_dev);
        BDF_HIP(hipGetLastError());
    }
    return BDF_OK;
}

// ---- relation-level side information (sample_beta_rel, src/sampling.jl:322-337) and alpha (sample_alpha, :129-134) -----
__global__ void k_rel_target(int64_t N, int64_t first_obs, const double *values, const double *pred, double inv_sqrt_alpha,
                             const double *alpha_dev, uint64_t seed, uint32_t sweep, uint32_t tag, double *v)
{
    if (alpha_dev) inv_sqrt_alpha = 1.0 / sqrt(*alpha_dev);          // (alpha sampled on the device: the same two IEEE operations as the host's)
    // v = (values - udot - mean) + alpha^-1/2 z,  pred = udot + mean; the noise is keyed by the observation's place in the relation
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) v[i] = (values[i] - pred[i]) + inv_sqrt_alpha * bdf_normal(seed, sweep, BDF_P_BETA_REL1, tag, (uint64_t)(first_obs + i), 0);
}

__global__ void k_rel_rhs(int64_t numF, double alpha, const double *alpha_dev, double lambda, uint64_t seed, uint32_t sweep, uint32_t tag,
                          double *rhs, double *rhs_scaled, double *lam_scaled)
{
    if (alpha_dev) alpha = *alpha_dev;
    // aFt_y = alpha F'v + sqrt(lambda) z;  the solve runs on (FF + (lambda / alpha) I) beta = aFt_y / alpha
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f < numF) {
        const double r = alpha * rhs[f] + sqrt(lambda) * bdf_normal(seed, sweep, BDF_P_BETA_REL2, tag, (uint64_t)f, 0);
        rhs[f] = r;
        rhs_scaled[f] = r / alpha;
    }
    if (f == 0) *lam_scaled = lambda / alpha;
}

__global__ void k_add_scalar(int64_t n, double a, double *x)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += a;
}

__global__ void k_sample_alpha(double lambda0, double nu0, double n, const double *sumsq, uint64_t seed, uint32_t sweep,
                               uint32_t tag, double *alpha_out)
{
    // Wishart(nu0 + n, SW) in one dimension: SW * chi2(nu0 + n) = SW * 2 Gamma((nu0 + n) / 2)
    const double SW = 1.0 / (1.0 / lambda0 + *sumsq);
    *alpha_out = SW * 2.0 * bdf_gamma(seed, sweep, tag, 0, 0.5 * (nu0 + n));
}

extern "C" int bdf_feat_linear(bdf_ctx *ctx, const bdf_feat *f, const double *beta, double mean_value, double *out)
{
    // out = mean_value + F beta (one column): linear_values of macau.jl:91, and the test rows' baseline of pred(r, probe, F)
    BDF_REQUIRE(ctx && f && beta && out, BDF_ERR_ARG, "bdf_feat_linear: NULL argument");
    int rc = feat_apply(ctx, f, false, beta, 1, f->n, 1, out, 1, f->m);
    if (rc) return rc;
    if (f->m > 0) {
        hipLaunchKernelGGL(k_add_scalar, dim3((unsigned)((f->m + 255) / 256)), dim3(256), 0, ctx->stream, f->m, mean_value, out);
        BDF_HIP(hipGetLastError());
    }
    return BDF_OK;
}

extern "C" int bdf_sample_alpha(bdf_ctx *ctx, double alpha_lambda0, double alpha_nu0, int64_t n, const double *sumsq_err,
                                uint32_t rel_tag, double *alpha_out)
{
    BDF_REQUIRE(ctx && sumsq_err && alpha_out, BDF_ERR_ARG, "bdf_sample_alpha: NULL argument");
    hipLaunchKernelGGL(k_sample_alpha, dim3(1), dim3(1), 0, ctx->stream, alpha_lambda0, alpha_nu0, (double)n, sumsq_err,
                       ctx->seed, ctx->sweep_host, 0x800000u | rel_tag, alpha_out);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

// several ranks: x (n doubles) := the sum of the ranks' x, block after block in rank order (every rank ends with the same
// bits); gb: world * n doubles of scratch
__global__ void k_sum_blocks(int64_t n, int world, const double *gb, double *x)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int r = 0; r < world; r++) s += gb[(size_t)r * n + i];
    x[i] = s;
}

static int sum_ranks_in(bdf_ctx *ctx, bdf_comm *comm, int rank, int world, double *x, int64_t n, double *gb)
{
    static const bool force = getenv("BDF_FORCE_COMM") != nullptr;      // (one rank through the collective all the same: tools/soak_determinism.py rccl)
    if ((world <= 1 && !(force && comm)) || n <= 0) return BDF_OK;
    int rc;
    BDF_HIP(hipMemcpyAsync(gb + (size_t)rank * n, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = bdf_allgather_block(ctx, comm, gb, (size_t)n * sizeof(double))) || (rc = bdf_allgather_join(ctx, comm))) return rc;
    hipLaunchKernelGGL(k_sum_blocks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, world, (const double *)gb, x);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

// (internal) the same with the caller's gather buffer (world * n doubles): for callers that hold the context's scratch themselves
int bdf_sum_ranks_into(bdf_ctx *ctx, bdf_comm *comm, double *x, int64_t n, double *gather)
{
    int rank = 0, world = 1, rc;
    if (comm && (rc = bdf_comm_size(comm, &rank, &world))) return rc;
    return sum_ranks_in(ctx, comm, rank, world, x, n, gather);
}

extern "C" int bdf_sum_ranks(bdf_ctx *ctx, bdf_comm *comm, double *x, int64_t n)
{
    BDF_REQUIRE(ctx && x && n >= 0, BDF_ERR_ARG, "bdf_sum_ranks: NULL argument or negative length");
    int rank = 0, world = 1, rc;
    if (comm && (rc = bdf_comm_size(comm, &rank, &world))) return rc;
    if (world <= 1) return BDF_OK;
    void *sv;
    if ((rc = bdf_scratch(ctx, (size_t)world * (size_t)n * sizeof(double), &sv))) return rc;
    return sum_ranks_in(ctx, comm, rank, world, x, n, (double *)sv);
}

extern "C" int bdf_sample_beta_rel(bdf_ctx *ctx, const bdf_feat *fc, const bdf_pairs *train, int D,
                                   const double *const *factors, double mean_value, double alpha, double lambda_beta,
                                   uint32_t rel_tag, double *beta_out, double *linear_out, double *rhs_out)
{
    return bdf_sample_beta_rel_ranks(ctx, nullptr, fc, train, 0, D, factors, mean_value, alpha, lambda_beta, rel_tag, beta_out,
                                     linear_out, rhs_out);
}

extern "C" int bdf_sample_beta_rel_ranks(bdf_ctx *ctx, bdf_comm *comm, const bdf_feat *fc, const bdf_pairs *train,
                                         int64_t first_obs, int D, const double *const *factors, double mean_value, double alpha,
                                         double lambda_beta, uint32_t rel_tag, double *beta_out, double *linear_out, double *rhs_out)
{
    return bdf_sample_beta_rel_impl(ctx, comm, fc, train, first_obs, D, factors, mean_value, alpha, nullptr, lambda_beta, rel_tag, beta_out,
                                    linear_out, rhs_out);
}

// (alpha_dev, nullable: the relation's precision in device memory -- sampled there inside bdf_gibbs_sweep -- instead of `alpha`)
int bdf_sample_beta_rel_impl(bdf_ctx *ctx, bdf_comm *comm, const bdf_feat *fc, const bdf_pairs *train, int64_t first_obs, int D,
                             const double *const *factors, double mean_value, double alpha, const double *alpha_dev, double lambda_beta,
                             uint32_t rel_tag, double *beta_out, double *linear_out, double *rhs_out)
{
    BDF_REQUIRE(ctx && fc && train && factors && beta_out && linear_out, BDF_ERR_ARG, "bdf_sample_beta_rel: NULL argument");
    BDF_REQUIRE((alpha_dev || alpha > 0.0) && lambda_beta >= 0.0, BDF_ERR_ARG, "bdf_sample_beta_rel: alpha must be positive, lambda_beta >= 0");
    BDF_REQUIRE(first_obs >= 0, BDF_ERR_ARG, "bdf_sample_beta_rel: first_obs must not be negative");
    int rank = 0, world = 1;
    if (comm) { int rcw = bdf_comm_size(comm, &rank, &world); if (rcw) return rcw; }
    bdf_feat *f = const_cast<bdf_feat *>(fc);
    const int64_t N = f->m, numF = f->n;
    BDF_REQUIRE(train->n == N, BDF_ERR_ARG,
                "bdf_sample_beta_rel: the relation has %lld observations but its feature matrix has %lld rows (DimensionMismatch)",
                (long long)train->n, (long long)N);
    const uint32_t tag = 0x800000u | rel_tag;
    // scratch (doubles): pred N | v N | t numF | rs numF | R P Z (numF each) | Tm N | scal 4 | lam 1, then ints
    // several ranks: this rank holds the rows [first_obs, first_obs + N) of the relation's feature matrix and the same
    // observations as pairs; F'v and (once) F'F are summed over the ranks in rank order, the solve is repeated on every rank
    const bool sum_ff = world > 1 && !f->FF_summed;
    const size_t gsz = world > 1 ? (size_t)world * (size_t)(sum_ff ? numF * numF : numF) : 0;
    const size_t total = 3 * (size_t)N + 5 * (size_t)numF + 16 + gsz;
    void *sv;
    int rc = bdf_scratch(ctx, total * sizeof(double) + 16 * sizeof(int), &sv);
    if (rc) return rc;
    double *pred = (double *)sv, *v = pred + N, *t = v + N, *rs = t + numF, *R = rs + numF, *P = R + numF, *Z = P + numF,
           *Tm = Z + numF, *scal = Tm + N, *lam = scal + 8, *gb = lam + 8;
    (void)R; (void)P; (void)Z; (void)Tm;
    if ((rc = bdf_predict_plain(ctx, train, D, factors, mean_value, pred))) return rc;
    if (N > 0) {
        hipLaunchKernelGGL(k_rel_target, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, N, first_obs,
                           (const double *)train->values_dev, (const double *)pred, 1.0 / sqrt(alpha), alpha_dev, ctx->seed,
                           ctx->sweep_host, tag, v);
        BDF_HIP(hipGetLastError());
    }
    if ((rc = feat_apply(ctx, f, true, v, 1, N, 1, t, 1, numF))) return rc;
    if ((rc = sum_ranks_in(ctx, comm, rank, world, t, numF, gb))) return rc;
    hipLaunchKernelGGL(k_rel_rhs, dim3((unsigned)((numF + 255) / 256)), dim3(256), 0, ctx->stream, numF, alpha, alpha_dev, lambda_beta,
                       ctx->seed, ctx->sweep_host, tag, t, rs, lam);
    BDF_HIP(hipGetLastError());
    if (rhs_out) BDF_HIP(hipMemcpyAsync(rhs_out, t, numF * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = ensure_FF(f))) return rc;
    if (sum_ff) {
        if ((rc = sum_ranks_in(ctx, comm, rank, world, f->FF_dev, numF * numF, gb))) return rc;
        f->FF_summed = true;
    }
    if (numF <= 16) hipLaunchKernelGGL(k_solve_small<16>, dim3(1), dim3(64), 0, ctx->stream, (int)numF, 1, f->FF_dev, lam, rs, beta_out, ctx->flag_dev);
    else if (numF <= 32) hipLaunchKernelGGL(k_solve_small<32>, dim3(1), dim3(64), 0, ctx->stream, (int)numF, 1, f->FF_dev, lam, rs, beta_out, ctx->flag_dev);
    else if (numF <= 64) hipLaunchKernelGGL(k_solve_small<64>, dim3(1), dim3(64), 0, ctx->stream, (int)numF, 1, f->FF_dev, lam, rs, beta_out, ctx->flag_dev);
    else if ((rc = bdf_chol_solve(ctx, f, 1, lam, rs, beta_out))) return rc;
    BDF_HIP(hipGetLastError());
    // linear_values = mean_value + F beta (macau.jl:91)
    if ((rc = feat_apply(ctx, f, false, beta_out, 1, numF, 1, linear_out, 1, N))) return rc;
    if (N > 0) {
        hipLaunchKernelGGL(k_add_scalar, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, N, mean_value, linear_out);
        BDF_HIP(hipGetLastError());
    }
    return BDF_OK;
}
