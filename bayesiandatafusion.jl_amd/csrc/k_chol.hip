// k_chol.hip -- solve_full of the reference (src/sampling.jl:314-320): beta = (FF + lambda I) \ rhs for 64 < numF <=
// compute_ff_size (6500, src/RelationData.jl:337-339), all D right-hand sides at once.
//
// Blocked right-looking Cholesky on v_mfma_f64_16x16x4_f64 tiles, block size 64:
//   W   : the trailing matrix (lower block-triangle) with the right-hand sides riding along as D extra ROWS (row NP + d holds
//         rhs[:, d]'), so that the forward solve L y = rhs is part of the factorisation: after step k the extra rows of block
//         column k hold (L^-1 rhs)' for that block.
//   step k, kernel 1 (one wave): the 64 x 64 diagonal block is factored (wave_linalg.h) and its inverse Inv_k = L_kk^-1 formed
//         (16 x 16 triangular inverses by substitution, the off-diagonal blocks by block recursion on the matrix cores).
//   step k, kernel 2 (one workgroup per pair of 64-row blocks below the diagonal block): P = W[:, k] Inv_k' (the panel, i.e.
//         column block k of L) for both row blocks, then W[bi, bj] -= P_bi P_bj'.  Workgroups of the first block column also
//         store the panel: transposed (LT) for the backward pass, and the rows of the right-hand sides (Y).
//   backward pass (one kernel, one workgroup per 16 right-hand sides): Z L = Y block column by block column from the last:
//         Z_k = (Y_k - sum_{j>k} Z_j L_jk) Inv_k; beta[:, d] = Z[d, :].
// Every operand of every MFMA is read so that the 16 lanes of a row group touch consecutive addresses.
#include "bdf_common.h"
#define BDF_CHOL_LOOKAHEAD        // a lone wave: the next pivot's reciprocal ahead of the step (c_layout_chol.h)
#include "c_layout_chol.h"
#include <algorithm>

namespace {

constexpr int CB = 64;
typedef double cd4 __attribute__((ext_vector_type(4)));

struct CholWork {
    int64_t n, NP, ld;      // numF, numF padded to 64, rows of W (NP + 64)
    int D, DP;              // right-hand sides, padded to 16
    double *W;              // ld x NP column-major
    double *LT;             // NP x NP: LT[c + r * NP] = L[r][c]
    double *Y;              // 64 x NP column-major: forward-solved right-hand sides as rows, then Z in place
    double *Inv, *InvT;     // NP / 64 blocks of 64 x 64 column-major: L_kk^-1 and its transpose
};

__global__ __launch_bounds__(256) void k_chol_setup(CholWork w, const double *__restrict__ FF, const double *__restrict__ lambda_p,
                                                    const double *__restrict__ rhs)
{
    const double lambda = *lambda_p;
    const int64_t total = w.ld * w.NP;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e % w.ld, j = e / w.ld;
        double v;
        if (i < w.NP) v = (i < w.n && j < w.n) ? FF[i + j * w.n] + (i == j ? lambda : 0.0) : (i == j ? 1.0 : 0.0);
        else { const int64_t d = i - w.NP; v = (d < w.D && j < w.n) ? rhs[j + d * w.n] : 0.0; }
        w.W[e] = v;
    }
}

// C (16 x 16, MFMA layout: lane (j = l & 15, h = l >> 4), register r = element (h + 4 r, j)) += A (16 x 16) B (16 x 16) with
// element accessors a(i, k), b(k, j)
template <typename FA, typename FB>
__device__ __forceinline__ void mma16(cd4 &c, int lane, FA a, FB b)
{
    const int i = lane & 15, h = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; t++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a(i, 4 * t + h), b(4 * t + h, i), c, 0, 0, 0);
}

// ---- step k, kernel 1: factor the diagonal block, invert it -----------------------------------------------------------------
__global__ __launch_bounds__(64) void k_chol_diag(CholWork w, int k, int *flag)
{
    using GG = Geo<CB>;
    __shared__ __attribute__((aligned(16))) double tri[GG::WAVE_LDS];
    __shared__ double sL[CB][CB + 1], sI[CB][CB + 1], sT[16][17], s_rs[CB];
    const int lane = threadIdx.x, j = lane & 15, h = lane >> 4;
    const int64_t kb = (int64_t)k * CB;
    // the block in the MFMA accumulator layout (register r of block (I, J) = element (16 I + h + 4 r, 16 J + j)), factored
    // in registers by the row sampler's factorisation (c_layout_chol.h); W holds the lower triangle
    double A[GG::NB * 4], bv[GG::DB], ts[GG::DB];
#pragma unroll
    for (int I = 0; I < GG::DB; I++)
#pragma unroll
        for (int J = 0; J <= I; J++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * I + h + 4 * r, colm = 16 * J + j;
                const int hi = row > colm ? row : colm, lo = row > colm ? colm : row;
                A[GG::blk(I, J) * 4 + r] = w.W[(kb + hi) + (kb + lo) * w.ld];
            }
#pragma unroll
    for (int J = 0; J < GG::DB; J++) { bv[J] = 0.0; ts[J] = 0.0; }
    // (the blocked variant: one wave alone factors a 64 x 64 block in 10 us instead of 16, c_layout_chol.h)
    factor_all_blocked<CB>(A, bv, ts, tri, j, h, CB, std::make_integer_sequence<int, CB - 1>{});
    wave_sync();
    // lane c = column c of the packed (unscaled) factor: Lt[i][c] = L[i][c] sqrt(d_c), Lt[c][c] = d_c
    const int c = lane;
    const typename GG::ColRT cr = GG::col_rt(c);
    double dv = tri[cr.cbase + (c & 3) * cr.nr4];
    if (!(dv > 0.0)) { atomicOr_system(flag, 8); dv = 1.0; }
    const double rs = fast_rsqrt(dv);
    for (int i = 0; i < CB; i++) {
        double v = 0.0;
        if (i > c) v = tri[cr.cbase + (i & 3) * cr.nr4 + (i >> 2) - cr.q] * rs;
        else if (i == c) v = dv * rs;
        sL[i][c] = v;
        sI[i][c] = 0.0;
    }
    wave_sync();
    // inverses of the four 16 x 16 diagonal blocks: lane (b, q) takes column q of inv(L_bb) by forward substitution
    {
        const int b = lane >> 4, q = lane & 15, o = 16 * b;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double s = (i == q) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < i; m++) s = fma(-sL[o + i][o + m], x[m], s);     // x[m] = 0 for m < q
            x[i] = (i >= q) ? s / sL[o + i][o + i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) sI[o + i][o + q] = x[i];
    }
    wave_sync();
    // off-diagonal blocks by block recursion: Inv_ij = -Inv_ii (sum_{m=j}^{i-1} L_im Inv_mj), block sub-diagonals in order
    for (int d = 1; d < 4; d++) {
        for (int bi = d; bi < 4; bi++) {
            const int bj = bi - d;
            cd4 t = cd4{0.0, 0.0, 0.0, 0.0};
            for (int m = bj; m < bi; m++)
                mma16(t, lane, [&](int i, int kk) { return sL[16 * bi + i][16 * m + kk]; },
                      [&](int kk, int j) { return sI[16 * m + kk][16 * bj + j]; });
#pragma unroll
            for (int r = 0; r < 4; r++) sT[(lane >> 4) + 4 * r][lane & 15] = t[r];
            wave_sync();
            cd4 u = cd4{0.0, 0.0, 0.0, 0.0};
            mma16(u, lane, [&](int i, int kk) { return sI[16 * bi + i][16 * bi + kk]; }, [&](int kk, int j) { return sT[kk][j]; });
            wave_sync();
#pragma unroll
            for (int r = 0; r < 4; r++) sI[16 * bi + (lane >> 4) + 4 * r][16 * bj + (lane & 15)] = -u[r];
            wave_sync();
        }
    }
    // Inv_k and its transpose (column-major 64 x 64 each); the diagonal block of L', for the record of the factor
    double *inv = w.Inv + (int64_t)k * CB * CB, *invt = w.InvT + (int64_t)k * CB * CB;
#pragma unroll 8
    for (int m = 0; m < CB; m++) {
        inv[lane + m * CB] = sI[lane][m];           // Inv[row = lane][col = m]
        invt[lane + m * CB] = sI[m][lane];          // InvT[row = lane][col = m] = Inv[m][lane]
        w.LT[(kb + lane) + (kb + m) * w.NP] = sL[m][lane];      // LT[c + r NP] = L[r][c], r = m, c = lane
    }
}

// ---- step k, kernel 2: panel (W[:, k] Inv_k') and trailing update for one pair of 64-row blocks ----------------------------
// blockIdx.x = bi, blockIdx.y = bj (blocks counted from the first row below the diagonal block; the last bi is the block of
// the right-hand sides); bj < 0 is encoded as gridDim.y == 1 && panel_only: store the panel, no update.
__global__ __launch_bounds__(256) void k_chol_update(CholWork w, int k, int panel_only)
{
    __shared__ double sP[2][CB][CB + 1];
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (!panel_only && bi < bj) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, h = lane >> 4;
    const int64_t kb = (int64_t)k * CB, r0 = kb + CB;
    const int nbm = (int)((w.NP - r0) / CB);                 // trailing matrix blocks; block nbm is the right-hand sides
    const double *invk = w.Inv + (int64_t)k * CB * CB;
    auto panel = [&](int b, int S) {
        // rows tile `wave` of block b: P (16 x 64) = W[rows, kb..kb+63] Inv_k'
        const int64_t row0 = (b < nbm ? r0 + (int64_t)b * CB : w.NP) + 16 * wave;
        const bool live = b < nbm || 16 * wave < w.DP;
        cd4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = cd4{0.0, 0.0, 0.0, 0.0};
        if (live) {
#pragma unroll 4
            for (int t = 0; t < 16; t++) {
                const int m = 4 * t + h;
                const double a = w.W[(row0 + i) + (kb + m) * w.ld];
#pragma unroll
                for (int q = 0; q < 4; q++)           // B[m][c] = Inv[c][m], c = 16 q + i
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, invk[(16 * q + i) + m * CB], acc[q], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) sP[S][16 * wave + h + 4 * r][16 * q + i] = acc[q][r];
    };
    panel(bi, 0);
    if (!panel_only && bj != bi) panel(bj, 1);
    __syncthreads();
    const int SJ = (!panel_only && bj != bi) ? 1 : 0;
    if (panel_only || bj == 0) {
        // store the panel: matrix rows transposed into LT, right-hand-side rows into Y
        if (bi < nbm) {
            for (int e = threadIdx.x; e < CB * CB; e += 256) {
                const int r = e / CB, c = e % CB;
                w.LT[(kb + c) + (r0 + (int64_t)bi * CB + r) * w.NP] = sP[0][r][c];
            }
        } else {
            for (int e = threadIdx.x; e < CB * CB; e += 256) {
                const int c = e / CB, d = e % CB;
                w.Y[d + (kb + c) * CB] = sP[0][d][c];
            }
        }
    }
    if (panel_only) return;
    // W[bi rows, bj cols] -= P_bi P_bj'; wave = row tile, four column tiles (the lower ones on the diagonal block)
    const int64_t rowt = (bi < nbm ? r0 + (int64_t)bi * CB : w.NP) + 16 * wave;
    if (bi == nbm && 16 * wave >= w.DP) return;
    for (int tj = 0; tj < 4; tj++) {
        if (bi == bj && tj > wave) break;
        const int64_t colt = r0 + (int64_t)bj * CB + 16 * tj;
        cd4 c4 = cd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int t = 0; t < 16; t++)
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(sP[0][16 * wave + i][4 * t + h], sP[SJ][16 * tj + i][4 * t + h], c4, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double *p = w.W + (rowt + h + 4 * r) + (colt + i) * w.ld;
            *p -= c4[r];
        }
    }
}

// ---- backward pass: Z L = Y, one workgroup per 16 right-hand sides, wave = 16-column chunk of a block ---------------------
__global__ __launch_bounds__(256) void k_chol_backward(CholWork w, double *__restrict__ beta)
{
    __shared__ double sA[16][CB + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, h = lane >> 4;
    const int d0 = 16 * blockIdx.x;
    const int nb = (int)(w.NP / CB);
    for (int k = nb - 1; k >= 0; k--) {
        const int64_t kb = (int64_t)k * CB, c0 = kb + 16 * wave;
        cd4 acc;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = w.Y[(d0 + h + 4 * r) + (c0 + i) * CB];
        cd4 sub = cd4{0.0, 0.0, 0.0, 0.0};
        for (int j = k + 1; j < nb; j++) {
            const int64_t jb = (int64_t)j * CB;
            // A[d][m] = Z[d][m] (written by this workgroup at step j: read past the L1), B[m][c] = L[m][c0 + c]; the block's 32
            // loads are issued before its 16 MFMAs
            double a[16], b[16];
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int64_t m = jb + 4 * t + h;
                a[t] = __hip_atomic_load(w.Y + (d0 + i) + m * CB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b[t] = w.LT[(c0 + i) + m * w.NP];
            }
#pragma unroll
            for (int t = 0; t < 16; t++) sub = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], sub, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) sA[h + 4 * r][16 * wave + i] = acc[r] - sub[r];
        __syncthreads();
        // Z_k[:, chunk] = A Inv_k[:, chunk]:  B[m][c] = Inv[m][16 wave + c] = InvT[(16 wave + c) + m * 64]
        const double *invt = w.InvT + (int64_t)k * CB * CB;
        cd4 z = cd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int t = 0; t < 16; t++) {
            const int m = 4 * t + h;
            z = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[i][m], invt[(16 * wave + i) + m * CB], z, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int d = d0 + h + 4 * r;
            const int64_t f = c0 + i;
            __hip_atomic_store(w.Y + d + f * CB, z[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d < w.D && f < w.n) beta[f + (int64_t)d * w.n] = z[r];
        }
        __threadfence();
        __syncthreads();
    }
}

}  // namespace

// beta (numF x D) = (FF + lambda I) \ rhs.  FF: dev numF x numF column-major (symmetric); lambda: dev scalar; rhs: dev numF x D.
// Workspace is kept on the feature object (allocated once).
int bdf_chol_solve(bdf_ctx *ctx, bdf_feat *f, int D, const double *lambda_dev, const double *rhs, double *beta)
{
    const int64_t n = f->n;
    CholWork w;
    w.n = n; w.NP = (n + CB - 1) / CB * CB; w.ld = w.NP + CB; w.D = D; w.DP = (D + 15) / 16 * 16;
    const int nb = (int)(w.NP / CB);
    const size_t nW = (size_t)w.ld * w.NP, nLT = (size_t)w.NP * w.NP, nY = (size_t)CB * w.NP, nI = (size_t)nb * CB * CB;
    const size_t total = nW + nLT + nY + 2 * nI;
    if (f->chol_ws_doubles < total) {
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        if (f->chol_ws) BDF_HIP(hipFree(f->chol_ws));
        f->chol_ws = nullptr; f->chol_ws_doubles = 0;
        BDF_HIP(hipMalloc((void **)&f->chol_ws, total * sizeof(double)));
        f->chol_ws_doubles = total;
    }
    w.W = f->chol_ws; w.LT = w.W + nW; w.Y = w.LT + nLT; w.Inv = w.Y + nY; w.InvT = w.Inv + nI;
    const unsigned sg = (unsigned)std::min<size_t>((nW + 255) / 256, 4096);
    hipLaunchKernelGGL(k_chol_setup, dim3(sg), dim3(256), 0, ctx->stream, w, (const double *)f->FF_dev, lambda_dev, rhs);
    for (int k = 0; k < nb; k++) {
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(64), 0, ctx->stream, w, k, ctx->flag_dev);
        const int nbm = nb - 1 - k;
        if (nbm == 0) hipLaunchKernelGGL(k_chol_update, dim3(1, 1), dim3(256), 0, ctx->stream, w, k, 1);
        else hipLaunchKernelGGL(k_chol_update, dim3(nbm + 1, nbm), dim3(256), 0, ctx->stream, w, k, 0);
    }
    hipLaunchKernelGGL(k_chol_backward, dim3(w.DP / 16), dim3(256), 0, ctx->stream, w, beta);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}
