// k_rows_lr.hip -- K1-lr: the latent rows with FEW observations, sampled without a D x D factorisation.
//
// Replaces, for the rows it takes, sample_user_basic (src/sampling.jl:200-212) of the reference: the same conditional
// distribution  x_i ~ N(inv(P_i) b_i, inv(P_i)),  P_i = Lambda + alpha W W',  b_i = Lambda mu + alpha W r  (W = the D x n
// gathered factor rows of the row's n observations, r = values - mean), drawn by ANOTHER map from standard normals than the
// reference's  chol(inv(P_i))' z + inv(P_i) b_i.  A row with n << D observations has P_i = Lambda + (rank n): the reference's
// map costs D^2 n + D^3/3 per row whatever n; this one costs ~2 D n^2 + n^3/3 + 2 D^2 (the last term as a plain dense product
// over all rows).  With Lambda = L L' (lower Cholesky, once per launch) and Vt = V L^-T (the opposite entity's whole factor
// matrix transformed once per launch: wt_o = L^-1 w_o is then a gathered row of Vt):
//
//     e0  = L' mu + u                      u, delta: D + n normals of the row's stream (BDF_P_ROW, entity_tag, original id):
//                                          u_d = number 2 (d % 16 + 16 (d / 32)) + (d / 16) % 2, delta_a = number
//                                          2 (16 ceil(ceil(D / 16) / 2) + a / 2) + a % 2 (the lanes that make them: k_rows_lr4)
//     G   = I_n + alpha Wt' Wt             n x n, on the matrix cores
//     tau = G^-1 (r - Wt' e0 - delta / sqrt(alpha))
//     q   = e0 + alpha Wt tau
//     x   = L^-T q                         (k_rowmat: a dense N x D x D product over the rows of the launch, in place)
//
// z = 0 gives x = mu + alpha Lambda^-1 W G^-1 (r - W' mu), the posterior mean in its Kalman-gain form; the noise part is the
// N(0, (I + Phi' Phi)^-1) sampler of Bhattacharya, Chakraborty & Mallick (2016) with Phi = sqrt(alpha) Wt'.  The oracle holds
// the same function (oracle/bdf_oracle.c: orc_sample_row_lowrank) and tests/test_oracle_known_answers.py proves -- the map
// is affine in z -- that its mean is inv(P_i) b_i and its S S' is inv(P_i) to 1e-10 for every n in 0 .. D/2 + 1.
// Parity: this kernel against that function on the same normals (1e-8), and >= 10^5-draw moments on the device.
//
// One wavefront per row (n <= 15).  The n rows of Vt arrive with the row kernel's coalesced gather (lane (j, h): element
// 16 I + j of observation 4 k + h), go to LDS once and come back in the MFMA operand layout with the contraction over D
// (lane (i, kk): elements kk D/4 .. of observation i); e0 rides along as "observation 15", which makes Wt' e0 column 15 of
// the same sixteen v_mfma_f64_16x16x4_f64.  G is then factored in the accumulator layout by the row kernel's own 16 x 16
// routines (c_layout_chol.h), the right-hand side riding along as the extra row.
#include "bdf_common.h"
#include "wave_linalg.h"
#include "c_layout_chol.h"
#include "dpp_rows16.h"
#include "dpp_rows32.h"

namespace {

template <int DP>
struct LrGeo {
    static constexpr int DB = DP / 16;
    static constexpr int KQ = DP / 4;                  // contraction elements per lane row of the MFMA operand (k_rowmat)
    // the rows go through LDS in pieces of HW = 32 elements (D = 64: two halves): 4.3 KB of staging per wave instead of 8.4,
    // six resident waves per SIMD instead of four
    static constexpr int HW = 32, NH = DP / HW, KH = HW / 4;
    static constexpr int LD = HW + 2;                  // doubles between staged rows: even (16-byte reads), odd multiple of two banks
    static constexpr int STAGE = 16 * LD;
    static constexpr int TRI = Geo<16>::WAVE_LDS;      // packed 16 x 16 factor (+ the extra row's panel); the normals sit there first
    static constexpr int WAVE_LDS = STAGE + TRI;
    static_assert(TRI >= DP + 16, "the D + n normals of a row fit the packed factor's space");
};

// ---- the launch's constants: L = chol(Lambda) (lower, natural order), then
//      Tf[d][c] = L^-T[d][c] = L^-1[c][d]   (Vt = V Tf: row m of Vt is L^-1 v_m)
//      Tb[d][c] = L^-1[d][c]                (x' = q' Tb:  x = L^-T q)
//      mt[d]    = (L' mu)[d]                 (shared prior mean; per-row prior means: Tm = L, and k_rowmat forms mu_i' L row by row)
// both matrices DP x DP row-major, zero outside D x D.  One wavefront; the matrix lives in LDS.
template <int DP>
__global__ __launch_bounds__(64) void k_lr_prep(int D, const double *Lambda, const double *mu,
                                                 double *__restrict__ Tf, double *__restrict__ Tb, double *__restrict__ Tm, double *__restrict__ mt,
                                                 double *__restrict__ zero_row, int *flag, const uint32_t *ready, uint32_t ready_want)
{
    constexpr int LDL = DP + 1;
    if (threadIdx.x < DP) zero_row[threadIdx.x] = 0.0;       // the row the padding lanes of k_rows_lr4 gather
    // launched without waiting for the hyperprior draw that writes (mu, Lambda) (bdf_gibbs_sweep on reserved CUs: the draw runs
    // on CUs the row stream never uses): poll its flag as the row kernel's waves do, then read past the non-coherent caches
    if (ready) {
        int spins = 0;
        while ((int32_t)(__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ready_want) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 22)) { if (threadIdx.x == 0) atomicOr_system(flag, 16); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // sA[i * LDL + c]: the Schur complement's lower triangle, then L in place; L^-1 (lower triangular too) goes TRANSPOSED
    // into the strict upper triangle -- X[i][c], i > c, at sA[c * LDL + i] -- and its diagonal into sD
    __shared__ double sA[DP * LDL];
    __shared__ double sD[DP];
    const int c = threadIdx.x;
    for (int e = c; e < DP * DP; e += 64) {
        const int i = e / DP, cc = e % DP;
        sA[i * LDL + cc] = (i < D && cc < D) ? __hip_atomic_load(Lambda + i + (int64_t)cc * D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : ((i == cc) ? 1.0 : 0.0);
    }
    wave_sync();
    bool bad = false;
    for (int k = 0; k < D; k++) {
        const double p = sA[k * LDL + k];
        if (!(p > 0.0)) bad = true;
        const double sd = sqrt(p);
        const double lck = (c < D && c > k) ? sA[c * LDL + k] / sd : 0.0;        // L[c][k]
        wave_sync();
        if (c == k) sA[k * LDL + k] = sd;
        else if (c < D && c > k) sA[c * LDL + k] = lck;
        wave_sync();
        // lane c updates ROW c of the trailing lower triangle: A[c][m] -= L[c][k] L[m][k], k < m <= c
        if (c < D && c > k)
            for (int m = k + 1; m <= c; m++) sA[c * LDL + m] = fma(-lck, sA[m * LDL + k], sA[c * LDL + m]);
        wave_sync();
    }
    if (bad && c == 0) atomicOr_system(flag, 1);
    // L^-1 by columns: lane c solves L x = e_c (x_i = 0 above the diagonal); its entries sit in row c of the upper triangle
    if (c < D) {
        const double xd = 1.0 / sA[c * LDL + c];
        sD[c] = xd;
        for (int i = c + 1; i < D; i++) {
            double s = -sA[i * LDL + c] * xd;
            for (int m = c + 1; m < i; m++) s = fma(-sA[i * LDL + m], sA[c * LDL + m], s);
            sA[c * LDL + i] = s / sA[i * LDL + i];
        }
    }
    wave_sync();
    for (int e = c; e < DP * DP; e += 64) {
        const int d = e / DP, cc = e % DP;
        const bool in = d < D && cc < D;
        // X[d][cc] (d >= cc) and X[cc][d] (cc >= d)
        const double lo = (d > cc) ? sA[cc * LDL + d] : ((d == cc && in) ? sD[d] : 0.0);
        const double up = (cc > d) ? sA[d * LDL + cc] : ((d == cc && in) ? sD[d] : 0.0);
        Tb[e] = in ? lo : 0.0;
        Tf[e] = in ? up : 0.0;
    }
    wave_sync();
    for (int e = c; e < DP * DP; e += 64) {            // Tm[d][cc] = L[d][cc]: (mu_i' L)[cc] = (L' mu_i)[cc] for per-row prior means
        const int d = e / DP, cc = e % DP;
        Tm[e] = (d < D && cc <= d) ? sA[d * LDL + cc] : 0.0;
    }
    if (c < DP && mu) {                                // (mu NULL: per-row prior means, transformed by k_rowmat)
        double s = 0.0;
        if (c < D)
            for (int i = c; i < D; i++) s = fma(sA[i * LDL + c], __hip_atomic_load(mu + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), s);
        mt[c] = s;
    }
}

// ---- Y = X T for rows of D doubles (X, Y row-major with leading dimension D; T: DP x DP row-major, zero-padded), on the
// matrix cores.  A tile is 16 rows; the DB = DP / 16 waves of a tile each take one 16-column block of the result and keep
// their block of T in registers; the contraction runs over d = kk DP/4 + s (lane row kk, k-step s).  rows == nullptr: rows
// 0 .. n_rows-1; else the listed rows (entries < 0: none).  In place (Y == X) is allowed: every wave of a tile has read the tile
// before any of them writes (workgroup barrier).
// The tile comes through LDS: its waves load it ONCE between them, 64 consecutive doubles per instruction (a lane reading the 128
// bytes of its own row's quarter -- the operand layout -- touched 64 cache lines per instruction, eight instructions per line, in
// every one of the tile's waves: configuration C4's back-transform of 9.75 M rows moved its 10 GB at 3 TB/s), and every wave reads
// its operands back from there; the next tile's loads are in flight under this tile's matrix instructions and stores.
template <int DP>
__global__ __launch_bounds__(256) void k_rowmat(const double *X, double *Y, const double *__restrict__ T, int D, int ldy, const int32_t *__restrict__ rows,
                                                int64_t n_rows, int64_t n_iters)
{
    constexpr int DB = DP / 16, KQ = DP / 4, TPW = 4 / DB, LDT = DP + 2;
    __shared__ __attribute__((aligned(16))) double tile[TPW][16 * LDT];
    __shared__ int64_t tile_rows[TPW][16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int sub = wave / DB, cb = wave % DB;
    double b[KQ];
#pragma unroll
    for (int s = 0; s < KQ; s++) b[s] = T[(kk * KQ + s) * DP + 16 * cb + i];
    // this wave's share of its tile's loads: rows cb * 16 / DB .. of the tile, 64 consecutive doubles per instruction
    int trow[4], tcol[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int E = cb * (16 / DB) * DP + q * 64 + lane;
        trow[q] = E / DP; tcol[q] = E % DP;
    }
    auto tile_row = [&](int64_t it, int r16) -> int64_t {
        const int64_t r = (it * TPW + sub) * 16 + r16;
        return (it < n_iters && r < n_rows) ? (rows ? (int64_t)rows[r] : r) : -1;
    };
    auto tile_load = [&](double (&g)[4], int64_t it) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t row = tile_row(it, trow[q]);
            g[q] = (row >= 0 && tcol[q] < D) ? X[row * D + tcol[q]] : 0.0;
        }
    };
    double g[4];
    tile_load(g, blockIdx.x);
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t myrow = tile_row(it, i);
#pragma unroll
        for (int q = 0; q < 4; q++) tile[sub][trow[q] * LDT + tcol[q]] = g[q];
        if (cb == 0 && kk == 0) tile_rows[sub][i] = myrow;
        __syncthreads();                               // the tile is in LDS: every global read of it has completed (in place: before any write)
        tile_load(g, it + gridDim.x);                  // the next tile's rows, under this tile's matrix instructions
        double a[KQ];
        const double *src = &tile[sub][i * LDT + kk * KQ];
#pragma unroll
        for (int s = 0; s < KQ; s += 2) { const d2 v = *(const d2 *)(src + s); a[s] = v[0]; a[s + 1] = v[1]; }
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KQ; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
        // C layout: lane (j = i, h = kk), register r: tile row h + 4 r, column 16 cb + j
        const int col = 16 * cb + i;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int64_t rowm = tile_rows[sub][kk + 4 * rr];
            if (rowm >= 0 && col < ldy) Y[rowm * ldy + col] = acc[rr];       // (columns D .. ldy-1: zeros, T is zero-padded)
        }
        __syncthreads();                               // every wave has read the tile from LDS: the next one may be written
    }
}

struct LrItem {
    int32_t row;          // where the sample is written (position in the factor matrix)
    int32_t orig;         // the row's original id (random stream)
    int64_t q_begin;
    int32_t count, _pad;
};

struct LrArgs {
    const int32_t *colidx;      // the relation's other-mode ids, mode order
    const double *vals;
    const double *vt;           // the opposite factor transformed: rows of DP doubles (zeros behind the first D), then one all-zero row
    int64_t zero_row;           // ... its index
    const double *mt;           // L' mu (mt_stride = 0), or row i's L' mu_i at mt + i * mt_stride (per-row prior means, macau.jl:104)
    int64_t mt_stride;
    double *out;
    double alpha, mean;
    const double *alpha_dev;    // nullable: the precision in device memory (else `alpha`)
    uint64_t seed;
    uint32_t sweep, entity_tag;
    int32_t D, _pad;
    int *flag;
};

template <int DP>
__global__ __launch_bounds__(256, 6) void k_rows_lr(LrArgs a, const LrItem *__restrict__ items, int64_t n_items)
{
    using LG = LrGeo<DP>;
    using G16 = Geo<16>;
    constexpr int DB = LG::DB, LD = LG::LD;
    __shared__ __attribute__((aligned(16))) double lds[4 * LG::WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    if (w >= n_items) return;
    double *st = lds + wave * LG::WAVE_LDS, *tri = st + LG::STAGE;
    const int j = lane & 15, h = lane >> 4;
    const LrItem it = items[w];
    if (it.row < 0) return;             // (the list is padded to a multiple of four with row = -1 records: k_rows_lr4's idle lane rows)
    const int D = a.D, n = it.count;
    const double alpha = a.alpha_dev ? *a.alpha_dev : a.alpha;

    // ---- the row's D + n normals, one Philox block and one Box-Muller pair per lane, through LDS (the packed factor's space)
    // (which numbers of the stream are u_d and delta_a: lr_u_index / lr_delta_index below -- the assignment follows the lanes of
    // the four-rows-per-wave kernel)
    const int zbase = 16 * (((D + 15) / 16 + 1) / 2);
    if (lane < zbase + (n + 1) / 2) {
        double z0, z1;
        bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, (uint32_t)lane, z0, z1);
        tri[2 * lane] = z0;
        tri[2 * lane + 1] = z1;
    }
    // ---- gather: lane (j, h) takes element 16 I + j of observations h, h + 4, h + 8, h + 12
    double wv[4][DB];
    double rv = 0.0;
    if (n > 0) {
        int64_t ix[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int o = 4 * k + h;
            ix[k] = a.colidx[it.q_begin + (o < n ? o : n - 1)];
        }
        if (j < n) rv = a.vals[it.q_begin + j] - a.mean;
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int I = 0; I < DB; I++) {
                const int e = 16 * I + j;
                const double v = a.vt[ix[k] * DP + e];
                wv[k][I] = (4 * k + h < n && e < D) ? v : 0.0;
            }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int I = 0; I < DB; I++) wv[k][I] = 0.0;
    }
    wave_sync();
    // e0 = L' mu + u in the lanes of lane row 3 (element 16 I + j), staged as "observation 15"; delta_j in lane j
    double e0[DB];
#pragma unroll
    for (int I = 0; I < DB; I++) {
        const int e = 16 * I + j;
        e0[I] = (e < D) ? a.mt[(int64_t)it.row * a.mt_stride + e] + tri[2 * (j + 16 * (I / 2)) + (I & 1)] : 0.0;
    }
    const double dl = (j < n) ? tri[2 * (zbase + (j >> 1)) + (j & 1)] : 0.0;
    // ---- G~ = S S' with S = [Wt' ; 0 ; e0'] (16 x D), HW columns at a time: operand lane (i = j, kk = h), k-step s:
    // S[i][HW hh + kk KH + s]
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int hh = 0; hh < LG::NH; hh++) {
        if (hh > 0) wave_sync();                      // the previous piece has been read
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int I2 = 0; I2 < 2; I2++)
                st[(4 * k + h) * LD + 16 * I2 + j] = (k == 3 && h == 3) ? e0[2 * hh + I2] : wv[k][2 * hh + I2];
        wave_sync();
        const double *src = st + j * LD + h * LG::KH;
        double x[LG::KH];
#pragma unroll
        for (int s = 0; s < LG::KH; s += 2) { const d2 v = *(const d2 *)(src + s); x[s] = v[0]; x[s + 1] = v[1]; }
#pragma unroll
        for (int s = 0; s < LG::KH; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s], x[s], acc, 0, 0, 0);
    }
    // ---- the n x n system G tau = rho in the accumulator layout (lane (j, h), register r: row h + 4 r, column j),
    // identity outside n x n; Wt' e0 is row 15 of the product
    const double g = __shfl(acc[3], 48 + j);
    double A[4], bv[1], ts[1];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int m = h + 4 * r;
        const double id = (m == j) ? 1.0 : 0.0;
        A[r] = (m < n && j < n) ? fma(alpha, acc[r], id) : id;
    }
    bv[0] = (j < n) ? rv - g - dl * fast_rsqrt(alpha) : 0.0;
    ts[0] = 0.0;
    const int Ds = n > 1 ? n : 1;
    wave_sync();                                       // the normals have been read: the space is the packed factor's now
    zero_packed_factor<16>(tri, lane);
    factor_all_blocked<16>(A, bv, ts, tri, j, h, Ds, std::make_integer_sequence<int, 15>{});
    const typename G16::ColRT cr = G16::col_rt(lane < 16 ? lane : 0);
    wave_sync();
    double dv = 1.0, tv = 0.0;
    if (lane < Ds) { dv = tri[cr.cbase + (lane & 3) * cr.nr4]; tv = ts[0]; }
    if (!(dv > 0.0)) atomicOr_system(a.flag, 1);
    const double rdv = fast_rcp(dv);
    double yh = tv;
    unsigned colq[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        colq[q] = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(tri + cr.cbase + q * cr.nr4 - cr.q);
    backward_all<16>(yh, rdv, colq, std::make_integer_sequence<int, 1>{});
    const double tau = (lane < n) ? yh * rdv : 0.0;
    // ---- q = e0 + alpha Wt tau: lane (j, h) adds its observations' parts, then the four lane rows are summed
    double t4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) t4[k] = __shfl(tau, 4 * k + h);
    double qv = 0.0;
#pragma unroll
    for (int I = 0; I < DB; I++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) s = fma(wv[k][I], t4[k], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const double v = fma(alpha, s, e0[I]);
        qv = (h == I) ? v : qv;
    }
    const int e = 16 * h + j;
    if (h < DB && e < D) a.out[(int64_t)it.row * D + e] = qv;
}


// ---- FOUR ROWS PER WAVE: every row of 16 lanes takes one entity row; lane b of it takes OBSERVATION b (n <= 16) -----------------
// The wave-per-row kernel above spends ~850 vector instructions and 16 matrix instructions on a row whatever its length; most
// of it (normals, the 16 x 16 factorisation, index arithmetic) is per-row overhead on a quarter-filled wave, and at D = 30 on
// an L2-resident factor that instruction stream is what bounds the launch.  Here lane b streams ITS observation's row of Vt
// through registers, eight elements at a time, and the Gram matrix grows by one rank-1 update per ELEMENT d -- G[., b] += wt_d[.]
// wt_d[b]: DR fmac_dpp with lane a of the lane row as the broadcast source, the same instruction k_rows_small uses per
// observation; Wt' e0 rides along as one more row.  The n x n system is then k_rows_small's column-per-lane LDL' with the
// right-hand side riding along, and q = e0 + alpha Wt tau takes a second, element-major read of the same rows (lane j:
// elements j, j + 16, ... of observation a, tau_a broadcast by DPP), which the caches hold.  ~2,000 vector instructions per
// FOUR rows at D = 64, ~1,300 at D <= 32.  Lane j makes pair j + 16 r of the row's stream -- u of elements 32 r + j and
// 32 r + 16 + j -- and pair DP / 2 + j / 2 for delta_j (orc_lowrank_normals is this assignment).
template <int S, int C>
__device__ __forceinline__ void lr4_load(double (&W)[2][8], const d2 *rowp)
{
#pragma unroll
    for (int u = 0; u < 4; u++) { const d2 v = rowp[4 * C + u]; W[S][2 * u] = v[0]; W[S][2 * u + 1] = v[1]; }
}
template <int DP, int DR, int C, int U>
__device__ __forceinline__ void lr4_gram_el(double (&A)[16], double &gacc, const double (&e0)[DP / 16], const double (&Wc)[8])
{
    if constexpr (U < 8) {
        const double w = Wc[U];
        small_rank1<DR, 0>(A, w);                                       // G[., b] += wt_d[.] wt_d[b]
        constexpr int d = 8 * C + U;
        fm1_run<d % 16>(gacc, e0[d / 16], w);                           // (Wt' e0)_b += e0_d wt_d[b]
        lr4_gram_el<DP, DR, C, U + 1>(A, gacc, e0, Wc);
    }
}
template <int DP, int DR, int C>
__device__ __forceinline__ void lr4_gram(double (&A)[16], double &gacc, const double (&e0)[DP / 16], double (&W)[2][8], const d2 *rowp)
{
    if constexpr (C < DP / 8) {
        lr4_gram_el<DP, DR, C, 0>(A, gacc, e0, W[C & 1]);
        if constexpr (C + 2 < DP / 8) lr4_load<(C & 1), C + 2>(W, rowp);
        lr4_gram<DP, DR, C + 1>(A, gacc, e0, W, rowp);
    }
}
template <int DP, int A0, int T>
__device__ __forceinline__ void lr4_cgather(double (&wc)[4][DP / 16], uint32_t idw, const double *vt, int j)
{
    if constexpr (T < 4) {
        const double *rp = vt + (int64_t)row_bcast_u32<A0 + T>(idw) * DP + j;
#pragma unroll
        for (int k = 0; k < DP / 16; k++) wc[T][k] = rp[16 * k];
        lr4_cgather<DP, A0, T + 1>(wc, idw, vt, j);
    }
}
template <int DP, int A0, int T>
__device__ __forceinline__ void lr4_cfma(double (&q)[DP / 16], const double (&wc)[4][DP / 16], double tau)
{
    if constexpr (T < 4) {
#pragma unroll
        for (int k = 0; k < DP / 16; k++) fm1_run<A0 + T>(q[k], tau, wc[T][k]);      // q_k += tau_a wt_a[16 k + j]
        lr4_cfma<DP, A0, T + 1>(q, wc, tau);
    }
}
template <int DP, int DR, int A0>
__device__ __forceinline__ void lr4_phase_c(double (&q)[DP / 16], uint32_t idw, const double *vt, int j, double tau)
{
    if constexpr (A0 < DR) {
        double wc[4][DP / 16];
        lr4_cgather<DP, A0, 0>(wc, idw, vt, j);
        lr4_cfma<DP, A0, 0>(q, wc, tau);
        lr4_phase_c<DP, DR, A0 + 4>(q, idw, vt, j, tau);
    }
}

template <int DP, int DR>
__device__ __forceinline__ void lr4_body(const LrArgs &a, const LrItem &it, const int j)
{
    constexpr int DB = DP / 16, NR = DP / 32;
    const bool live = it.row >= 0;
    const int D = a.D, n = live ? it.count : 0;
    const double alpha = a.alpha_dev ? *a.alpha_dev : a.alpha;
    uint32_t idw = (uint32_t)a.zero_row;
    double rv = 0.0;
    if (j < n) {
        idw = (uint32_t)a.colidx[it.q_begin + j];
        rv = a.vals[it.q_begin + j] - a.mean;
    }
    const d2 *rowp = (const d2 *)(a.vt + (int64_t)idw * DP);
    double W[2][8];
    lr4_load<0, 0>(W, rowp);
    lr4_load<1, 1>(W, rowp);
    // the normals under the first loads
    double e0[DB];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        double z0, z1;
        bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, (uint32_t)(j + 16 * r), z0, z1);
        const double *mrow = a.mt + (live ? (int64_t)it.row * a.mt_stride : 0);
        e0[2 * r] = mrow[32 * r + j] + z0;
        e0[2 * r + 1] = mrow[32 * r + 16 + j] + z1;
    }
    double dl;
    {
        double z0, z1;
        bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, (uint32_t)(DP / 2 + (j >> 1)), z0, z1);
        dl = (j & 1) ? z1 : z0;
    }
    double A[16];
#pragma unroll
    for (int i = 0; i < 16; i++) A[i] = 0.0;
    double gacc = 0.0;
    asm volatile("s_nop 1" ::: "memory");              // (e0 is a DPP source below: written by the vector adds just above)
    lr4_gram<DP, DR, 0>(A, gacc, e0, W, rowp);
    // G = I + alpha Wt' Wt on n x n, identity outside; rho = r - Wt' e0 - delta / sqrt(alpha)
#pragma unroll
    for (int i = 0; i < DR; i++) {
        const double id = (i == j) ? 1.0 : 0.0;
        A[i] = (i < n && j < n) ? fma(alpha, A[i], id) : id;
    }
    double rho = (j < n) ? rv - gacc - dl * fast_rsqrt(alpha) : 0.0;
    double dj = 1.0;
    small_factor<DR, 0>(A, rho, dj, j);
    if (j < n && !(dj > 0.0)) atomicOr_system(a.flag, 1);
    const double rdj = fast_rcp(dj);
    double tau = rho * rdj;
    small_backward<DR - 1>(A, tau, rdj, j);
    // q = e0 + alpha Wt tau: the rows once more, element-major (the padding's observations are the zero row, their tau is 0)
    double q[DB];
#pragma unroll
    for (int k = 0; k < DB; k++) q[k] = 0.0;
    asm volatile("s_nop 1" ::: "memory");              // (tau is a DPP source below)
    lr4_phase_c<DP, DR, 0>(q, idw, a.vt, j, tau);
    if (live) {
#pragma unroll
        for (int k = 0; k < DB; k++)
            if (16 * k + j < D) a.out[(int64_t)it.row * D + 16 * k + j] = fma(alpha, q[k], e0[k]);
    }
}

template <int DP>
__global__ __launch_bounds__(256, 3) void k_rows_lr4(LrArgs a, const LrItem *__restrict__ items, int64_t n_items)
{
    const int lane = threadIdx.x & 63, j = lane & 15;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w * 4 >= n_items) return;
    const LrItem it = items[w * 4 + (lane >> 4)];
    int nmax = it.row >= 0 ? it.count : 0;
    nmax = max(nmax, __shfl_xor(nmax, 16));
    nmax = max(nmax, __shfl_xor(nmax, 32));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    if (nmax <= 4) lr4_body<DP, 4>(a, it, j);
    else if (nmax <= 8) lr4_body<DP, 8>(a, it, j);
    else if (nmax <= 12) lr4_body<DP, 12>(a, it, j);
    else lr4_body<DP, 16>(a, it, j);
}

// ---- ROWS OF 17 .. 32 OBSERVATIONS (D > 32): four rows per wave still, a lane row per entity row, lane b of it taking observations b
// AND 16 + b -- the 32 x 32 Gram system in K1c's layout (dpp_rows32.h: lane b holds columns b and 16 + b).  The two rows of Vt stream
// through registers eight elements at a time; every ELEMENT d is one rank-1 update of the three stored blocks (col_rank1: 48 fmac_dpp)
// and of Wt' e0 (the system's extra row); block (0,1) is block (1,0)'s transpose through LDS; G = I + alpha Wt' Wt, rho rides along
// through K1c's LDL' (fin_factor) and backward solve (fin_backward: tau); q = e0 + alpha Wt tau takes the element-major second read.
// Normals: u as k_rows_lr4; delta_a = pair DP / 2 + a / 2, element a % 2 (orc_lowrank_normals' assignment for every a).
template <int DP, int C, int U>
__device__ __forceinline__ void lr32_gram_el(double (&A0)[33], double (&A1)[33], const double (&e0)[DP / 16], const double (&W0c)[8], const double (&W1c)[8])
{
    if constexpr (U < 8) {
        const double w0 = W0c[U], w1 = W1c[U];
        col_rank1<32, 0>(A0, A1, w0, w1);                                // G[., b] += wt_d[.] wt_d[b], G[., 16 + b] += wt_d[.] wt_d[16 + b]
        constexpr int d = 8 * C + U;
        fm1_run<d % 16>(A0[32], e0[d / 16], w0);                         // (Wt' e0)_b      += e0_d wt_d[b]
        fm1_run<d % 16>(A1[32], e0[d / 16], w1);                         // (Wt' e0)_(16+b) += e0_d wt_d[16 + b]
        lr32_gram_el<DP, C, U + 1>(A0, A1, e0, W0c, W1c);
    }
}
template <int DP, int C>
__device__ __forceinline__ void lr32_gram(double (&A0)[33], double (&A1)[33], const double (&e0)[DP / 16], double (&W0)[2][8], double (&W1)[2][8],
                                          const d2 *rowp0, const d2 *rowp1)
{
    if constexpr (C < DP / 8) {
        lr32_gram_el<DP, C, 0>(A0, A1, e0, W0[C & 1], W1[C & 1]);
        if constexpr (C + 2 < DP / 8) { lr4_load<(C & 1), C + 2>(W0, rowp0); lr4_load<(C & 1), C + 2>(W1, rowp1); }
        lr32_gram<DP, C + 1>(A0, A1, e0, W0, W1, rowp0, rowp1);
    }
}
template <int DP, int A0I>
__device__ __forceinline__ void lr32_phase_c(double (&q)[DP / 16], uint32_t idw0, uint32_t idw1, const double *vt, int j, double tau0, double tau1)
{
    if constexpr (A0I < 32) {
        double wc[4][DP / 16];
        if constexpr (A0I < 16) { lr4_cgather<DP, A0I, 0>(wc, idw0, vt, j); lr4_cfma<DP, A0I, 0>(q, wc, tau0); }
        else { lr4_cgather<DP, A0I - 16, 0>(wc, idw1, vt, j); lr4_cfma<DP, A0I - 16, 0>(q, wc, tau1); }
        lr32_phase_c<DP, A0I + 4>(q, idw0, idw1, vt, j, tau0, tau1);
    }
}

template <int DP>
__global__ __launch_bounds__(256, 2) void k_rows_lr32(LrArgs a, const LrItem *__restrict__ items, int64_t n_items)
{
    constexpr int DB = DP / 16, NR = DP / 32;
    __shared__ double lds[4 * 4 * 272];                   // block (1,0) of the four systems of each wave on its way to block (0,1)
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    if (w * 4 >= n_items) return;
    const LrItem it = items[w * 4 + g];
    const bool live = it.row >= 0;
    const int D = a.D, n = live ? it.count : 0;
    const double alpha = a.alpha_dev ? *a.alpha_dev : a.alpha;
    uint32_t idw0 = (uint32_t)a.zero_row, idw1 = (uint32_t)a.zero_row;
    double rv0 = 0.0, rv1 = 0.0;
    if (j < n) { idw0 = (uint32_t)a.colidx[it.q_begin + j]; rv0 = a.vals[it.q_begin + j] - a.mean; }
    if (16 + j < n) { idw1 = (uint32_t)a.colidx[it.q_begin + 16 + j]; rv1 = a.vals[it.q_begin + 16 + j] - a.mean; }
    const d2 *rowp0 = (const d2 *)(a.vt + (int64_t)idw0 * DP), *rowp1 = (const d2 *)(a.vt + (int64_t)idw1 * DP);
    double W0[2][8], W1[2][8];
    lr4_load<0, 0>(W0, rowp0); lr4_load<0, 0>(W1, rowp1);
    lr4_load<1, 1>(W0, rowp0); lr4_load<1, 1>(W1, rowp1);
    // the normals under the first loads
    double e0[DB];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        double z0, z1;
        bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, (uint32_t)(j + 16 * r), z0, z1);
        const double *mrow = a.mt + (live ? (int64_t)it.row * a.mt_stride : 0);
        e0[2 * r] = mrow[32 * r + j] + z0;
        e0[2 * r + 1] = mrow[32 * r + 16 + j] + z1;
    }
    double dl0, dl1;
    {
        double z0, z1;
        bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, (uint32_t)(DP / 2 + (j >> 1)), z0, z1);
        dl0 = (j & 1) ? z1 : z0;
        bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, (uint32_t)(DP / 2 + 8 + (j >> 1)), z0, z1);
        dl1 = (j & 1) ? z1 : z0;
    }
    double A0[33], A1[33];
#pragma unroll
    for (int i = 0; i < 33; i++) { A0[i] = 0.0; A1[i] = 0.0; }
    asm volatile("s_nop 1" ::: "memory");              // (e0 is a DPP source below: written by the vector adds just above)
    lr32_gram<DP, 0>(A0, A1, e0, W0, W1, rowp0, rowp1);
    // block (0,1) = block (1,0)': lane j's entry (i, 16 + j) is lane i's entry (16 + j, i)
    {
        double *tl = lds + (wave * 4 + g) * 272;
#pragma unroll
        for (int r = 0; r < 16; r++) tl[r * 17 + j] = A0[16 + r];
        wave_sync();
#pragma unroll
        for (int i = 0; i < 16; i++) A1[i] = tl[j * 17 + i];
        wave_sync();
    }
    // G = I + alpha Wt' Wt on n x n, identity outside; rho = r - Wt' e0 - delta / sqrt(alpha) as the extra row
    const double rsa = fast_rsqrt(alpha);
    const double rho0 = (j < n) ? rv0 - A0[32] - dl0 * rsa : 0.0, rho1 = (16 + j < n) ? rv1 - A1[32] - dl1 * rsa : 0.0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const double id0 = (i == j) ? 1.0 : 0.0, id1 = (i == 16 + j) ? 1.0 : 0.0;
        A0[i] = (i < n && j < n) ? fma(alpha, A0[i], id0) : id0;
        A1[i] = (i < n && 16 + j < n) ? fma(alpha, A1[i], id1) : id1;
    }
    A0[32] = rho0; A1[32] = rho1;
    asm volatile("s_nop 1" ::: "memory");              // (the matrix rows are DPP sources in the factorisation)
    double d0 = 1.0, d1 = 1.0;
    fin_factor<32, 0>(A0, A1, d0, d1, j);
    if (live && ((j < n && !(d0 > 0.0)) || (16 + j < n && !(d1 > 0.0)))) atomicOr_system(a.flag, 1);
    const double rd0 = fast_rcp(d0), rd1 = fast_rcp(d1);
    double tau0 = A0[32] * rd0, tau1 = A1[32] * rd1;
    fin_backward<31>(A0, A1, tau0, tau1, rd0, rd1, j);
    // q = e0 + alpha Wt tau: the rows once more, element-major (the padding's observations are the zero row, their tau is 0)
    double q[DB];
#pragma unroll
    for (int k = 0; k < DB; k++) q[k] = 0.0;
    asm volatile("s_nop 1" ::: "memory");              // (tau is a DPP source below)
    lr32_phase_c<DP, 0>(q, idw0, idw1, a.vt, j, tau0, tau1);
    if (live) {
#pragma unroll
        for (int k = 0; k < DB; k++)
            if (16 * k + j < D) a.out[(int64_t)it.row * D + 16 * k + j] = fma(alpha, q[k], e0[k]);
    }
}

int lr_buffers(bdf_ctx *ctx, size_t vt_bytes, size_t mrows_bytes)
{
    if (!ctx->lr_T) BDF_HIP(hipMalloc((void **)&ctx->lr_T, (size_t)(3 * 64 * 64 + 64) * sizeof(double)));
    if (mrows_bytes > ctx->lr_mrows_bytes) {
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->lr_mrows) BDF_HIP(hipFree(ctx->lr_mrows));
        ctx->lr_mrows = nullptr; ctx->lr_mrows_bytes = 0;
        const size_t nb = (mrows_bytes + 255) & ~(size_t)255;
        BDF_HIP(hipMalloc((void **)&ctx->lr_mrows, nb));
        ctx->lr_mrows_bytes = nb;
    }
    if (vt_bytes > ctx->lr_vt_bytes) {
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->lr_vt) BDF_HIP(hipFree(ctx->lr_vt));
        ctx->lr_vt = nullptr; ctx->lr_vt_bytes = 0;
        const size_t nb = (vt_bytes + 255) & ~(size_t)255;
        BDF_HIP(hipMalloc((void **)&ctx->lr_vt, nb));
        ctx->lr_vt_bytes = nb;
    }
    return BDF_OK;
}

template <int DP>
int lr_launch_t(bdf_ctx *ctx, const SampleArgs &a, int64_t M_other, const void *items, int64_t n_items, int64_t n_padded, int64_t n32_padded,
                const int32_t *rows_dev, bool transform, hipEvent_t e0, hipEvent_t e1)
{
    const int D = a.D;
    double *Tf = ctx->lr_T, *Tb = Tf + 64 * 64, *Tm = Tb + 64 * 64, *mt = Tm + 64 * 64;
    constexpr int TPW = 4 / (DP / 16);
    if (transform) {
        // (k_lr_prep also zeroes row M_other of the transformed matrix: what lanes without an observation gather)
        hipExtLaunchKernelGGL((k_lr_prep<DP>), dim3(1), dim3(64), 0, ctx->stream, e0, nullptr, 0, D, a.Lambda, a.mu_is_matrix ? (const double *)nullptr : a.mu,
                              Tf, Tb, Tm, mt, ctx->lr_vt + M_other * DP, a.flag, a.ready, a.ready_want);
        e0 = nullptr;
        const int64_t iters = (M_other + 16 * TPW - 1) / (16 * TPW);
        hipLaunchKernelGGL((k_rowmat<DP>), dim3((unsigned)std::min<int64_t>(iters, 4096)), dim3(256), 0, ctx->stream, a.t[0].fac[0], ctx->lr_vt,
                           (const double *)Tf, D, DP, (const int32_t *)nullptr, M_other, iters);
    }
    if (a.mu_is_matrix) {
        // per-row prior means (entity side information, macau.jl:103-104): L' mu_i for the rows of this launch, (mu_i' L) row by row
        const int64_t it2 = (n_items + 16 * TPW - 1) / (16 * TPW);
        hipLaunchKernelGGL((k_rowmat<DP>), dim3((unsigned)std::min<int64_t>(it2, 4096)), dim3(256), 0, ctx->stream, a.mu, ctx->lr_mrows,
                           (const double *)Tm, D, DP, rows_dev, n_items, it2);
    }
    LrArgs la;
    la.colidx = a.t[0].colidx; la.vals = a.t[0].vals; la.vt = ctx->lr_vt; la.zero_row = M_other; la.out = a.out;
    la.mt = a.mu_is_matrix ? ctx->lr_mrows : mt; la.mt_stride = a.mu_is_matrix ? DP : 0;
    la.alpha = a.t[0].alpha; la.alpha_dev = a.t[0].alpha_dev; la.mean = a.t[0].mean; la.seed = a.seed; la.sweep = a.sweep; la.entity_tag = a.entity_tag;
    la.D = D; la._pad = 0; la.flag = a.flag;
    static const bool wave_per_row = getenv("BDF_LR_WAVE") != nullptr;        // the wave-per-row kernel instead (rows of at most 15 observations)
    // items: n_padded records of rows of at most 16 observations, then n32_padded of rows of 17 .. 32 (D > 32 only); n_items rows in all
    if (n_padded > 0) {
        if (wave_per_row)
            hipExtLaunchKernelGGL((k_rows_lr<DP>), dim3((unsigned)((n_padded + 3) / 4)), dim3(256), 0, ctx->stream, e0, nullptr, 0, la, (const LrItem *)items, n_padded);
        else
            hipExtLaunchKernelGGL((k_rows_lr4<DP>), dim3((unsigned)((n_padded + 15) / 16)), dim3(256), 0, ctx->stream, e0, nullptr, 0, la, (const LrItem *)items, n_padded);
        e0 = nullptr;
    }
    if constexpr (DP == 64) {
        if (n32_padded > 0)
            hipExtLaunchKernelGGL((k_rows_lr32<DP>), dim3((unsigned)((n32_padded + 15) / 16)), dim3(256), 0, ctx->stream, e0, nullptr, 0, la,
                                  (const LrItem *)items + n_padded, n32_padded);
    }
    const int64_t iters = (n_items + 16 * TPW - 1) / (16 * TPW);
    hipExtLaunchKernelGGL((k_rowmat<DP>), dim3((unsigned)std::min<int64_t>(iters, 4096)), dim3(256), 0, ctx->stream, nullptr, e1, 0, (const double *)a.out, a.out,
                          (const double *)Tb, D, D, rows_dev, n_items, iters);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

}  // namespace

// The rows `items` (LrItem: one two-mode relation, at most 16 observations each, shared or per-row prior means (n_rows_entity rows
// of the factor matrix); n_items of them, padded
// with row = -1 records to n_padded, a multiple of four) of the launch described by `a`.  transform: L = chol(Lambda), the
// opposite factor's M_other rows transformed into the context's buffer (false: both are still valid from the previous chunk
// of the same entity launch).  rows_dev: the rows' positions (n_items int32).
int bdf_lr_launch(bdf_ctx *ctx, const SampleArgs &a, int64_t M_other, int64_t n_rows_entity, const void *items, int64_t n_items, int64_t n_padded,
                  int64_t n32_padded, const int32_t *rows_dev, bool transform, hipEvent_t e0, hipEvent_t e1)
{
    const int DP = a.D <= 32 ? 32 : 64;
    int rc = lr_buffers(ctx, ((size_t)M_other + 2) * DP * sizeof(double),         // rows of DP doubles, the zero row, slack
                        a.mu_is_matrix ? (size_t)n_rows_entity * DP * sizeof(double) : 0);
    if (rc) return rc;
    if (DP == 32) return lr_launch_t<32>(ctx, a, M_other, items, n_items, n_padded, 0, rows_dev, transform, e0, e1);
    return lr_launch_t<64>(ctx, a, M_other, items, n_items, n_padded, n32_padded, rows_dev, transform, e0, e1);
}

int bdf_lr_max_observations() { return getenv("BDF_LR_WAVE") ? 15 : 16; }
// ... and with the two-observations-per-lane kernel (k_rows_lr32, D > 32)
int bdf_lr32_max_observations() { return getenv("BDF_LR_WAVE") || (getenv("BDF_LR32") && atoi(getenv("BDF_LR32")) == 0) ? 0 : 32; }
