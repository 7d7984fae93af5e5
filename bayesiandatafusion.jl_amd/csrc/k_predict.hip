// k_predict.hip -- K7: test-set prediction and the running posterior mean of macau.jl:142-203.
//
// pred(r, test_vec) = udot(r, test_vec) + mean_value (src/sampling.jl:9-14); udot is the sum over the latent
// dimension of the product of the modes' factor rows (src/sampling.jl:30-45).  8 lanes share one test pair and read
// 32 bytes each, so that a gathered factor row is read as whole 128-byte segments.
#include "bdf_common.h"
#include <algorithm>
#include <cstdlib>

namespace {


struct PredArgs {
    int D, n_modes;
    int64_t n;
    const int32_t *ids;            // n_modes planes of n, 0-based
    const double *fac[BDF_MAX_MODES];
    double mean;
    const double *linear;          // nullable: per-pair baseline instead of mean (relation features: linear_values)
    const int32_t *orig;           // nullable: the pairs are stored sorted; orig[pair] = the caller's index (out, linear)
    int sorted_mode;               // the mode they are sorted by (-1: none)
    const double *values;
    double *out;                   // nullable: raw predictions
    double *avg, *sq;              // running state (update mode)
    int phase;                     // -1: predict only
    double count, clamp_lo, clamp_hi, cut;
    double *stats;
    double *partial;               // per-block statistics
};

__device__ inline double clampv(double x, double lo, double hi)
{
    if (lo > hi) return x;
    return x < lo ? lo : (x > hi ? hi : x);
}

// ---- what a lane does for the pair it owns: everything of macau.jl:142-184 that is not the gather -----------------------
// A group of 8 lanes computes the dot products of 8 consecutive pairs together (32 bytes of a factor row per lane) and then
// lane `sub` owns pair p0 + sub: ids, value and running state are read 8 consecutive pairs per group and instruction before
// the first gather is issued, and written back the same way.  (One lane per group doing the updates one after the other
// issued six times as many memory instructions as the gather itself, each with 8 active lanes 128 B apart.)
struct PairState {
    int64_t pm, po;                // storage position; the caller's index (out, linear)
    bool ok;
    double y, av, sv, base;
};

__device__ inline void pair_load(const PredArgs &a, int64_t p, PairState &s)
{
    s.ok = p < a.n;
    s.pm = s.ok ? p : a.n - 1;
    s.po = a.orig ? (int64_t)a.orig[s.pm] : s.pm;
    s.base = a.linear ? a.linear[s.po] : a.mean;
    s.y = a.phase >= 0 ? a.values[s.pm] : 0.0;
    s.av = 0.0; s.sv = 0.0;
    if (a.phase == 2) { s.av = a.avg[s.pm]; s.sv = a.sq[s.pm]; }
}

__device__ inline void pair_finish(const PredArgs &a, const PairState &s, double dot, double (&st)[4])
{
    if (!s.ok) return;
    const double p = dot + s.base;
    if (a.out) a.out[s.po] = p;
    if (a.phase >= 0) {
        double avg;
        if (a.phase == 0 || a.phase == 3) { avg = p; }
        else if (a.phase == 1) { avg = p; a.sq[s.pm] = p * p; }
        else { avg = (a.count * s.av + p) / (a.count + 1.0); a.sq[s.pm] = s.sv + p * p; }
        if (a.phase != 3) a.avg[s.pm] = avg;           // phase 3: statistics of this sample only, no running state
        const double ea = s.y - clampv(avg, a.clamp_lo, a.clamp_hi), ep = s.y - clampv(p, a.clamp_lo, a.clamp_hi);
        const bool label = s.y < a.cut;
        st[0] += ea * ea; st[1] += ep * ep;
        st[2] += (label == (avg < a.cut)) ? 1.0 : 0.0;
        st[3] += (label == (p < a.cut)) ? 1.0 : 0.0;
    }
}

__device__ inline void block_stats(const PredArgs &a, const double (&st)[4])
{
    __shared__ double red[4][256 / 64];
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double v = st[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[q][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 4) a.partial[blockIdx.x * 4 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

// General kernel: NM modes, any order of the pairs.  VEC = 4: D a multiple of 4, NC 32-byte pieces of a row per lane
// (D <= 32: one, D <= 64: two); VEC = 1: any D, a lane takes elements sub, sub + 8, ...  A trip is 8 consecutive pairs, their
// rows gathered BATCH pairs at a time.
template <int NM, int VEC, int NC>
__global__ __launch_bounds__(256) void k_predict(PredArgs a)
{
    const int tid = threadIdx.x, sub = tid & 7;
    double st[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t ngroups = (int64_t)gridDim.x * 32, ntrips = (a.n + 7) / 8;
    constexpr int BATCH = (VEC == 1) ? 2 : (NC * NM <= 3 ? 4 : 2);
    for (int64_t trip = (int64_t)blockIdx.x * 32 + tid / 8; trip < ntrips; trip += ngroups) {
        const int64_t p0 = trip * 8;
        PairState ps;
        pair_load(a, p0 + sub, ps);
        int32_t my[NM];
#pragma unroll
        for (int k = 0; k < NM; k++) my[k] = a.ids[(int64_t)k * a.n + ps.pm];
        double keep = 0.0;
#pragma unroll
        for (int u0 = 0; u0 < 8; u0 += BATCH) {
            if (p0 + u0 >= a.n) break;                     // group-uniform
            double s[BATCH];
            if constexpr (VEC == 4) {
                double4 f[BATCH][NM][NC];
#pragma unroll
                for (int u = 0; u < BATCH; u++)
#pragma unroll
                    for (int k = 0; k < NM; k++) {
                        const double *row = a.fac[k] + (int64_t)__shfl(my[k], u0 + u, 8) * a.D;
#pragma unroll
                        for (int c = 0; c < NC; c++) {
                            const int e = sub * 4 + 32 * c;
                            f[u][k][c] = e < a.D ? *(const double4 *)(row + e) : double4{0.0, 0.0, 0.0, 0.0};
                        }
                    }
#pragma unroll
                for (int u = 0; u < BATCH; u++) {
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        double4 p = f[u][0][c];
#pragma unroll
                        for (int k = 1; k < NM; k++) { p.x *= f[u][k][c].x; p.y *= f[u][k][c].y; p.z *= f[u][k][c].z; p.w *= f[u][k][c].w; }
                        if (sub * 4 + 32 * c < a.D) acc += (p.x + p.y) + (p.z + p.w);
                    }
                    s[u] = acc;
                }
            } else {
#pragma unroll
                for (int u = 0; u < BATCH; u++) {
                    const double *row[NM];
#pragma unroll
                    for (int k = 0; k < NM; k++) row[k] = a.fac[k] + (int64_t)__shfl(my[k], u0 + u, 8) * a.D;
                    double acc = 0.0;
                    for (int e = sub; e < a.D; e += 8) {
                        double p = 1.0;
#pragma unroll
                        for (int k = 0; k < NM; k++) p *= row[k][e];
                        acc += p;
                    }
                    s[u] = acc;
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                double v = s[u];
                v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
                if (sub == u0 + u) keep = v;
            }
        }
        pair_finish(a, ps, keep, st);
    }
    if (a.phase >= 0) block_stats(a, st);
}

// Pairs stored sorted by one mode (bdf_pairs_sort), two-mode relation, D a multiple of 4 up to 32: a group of 8 lanes walks RUN
// consecutive pairs and keeps the factor row of the sorted mode in registers while its id does not change -- the update then
// gathers one row per pair instead of two (MovieLens test set sorted by movie: 126 pairs per row; 136 MB instead of 256 MB
// of L2 gathers per update, which is what the update and the row kernel running beside it compete for).  A lane owns two
// pairs of the run (p0 + sub, p0 + 8 + sub), see PairState.
constexpr int RUN = 16;
__global__ __launch_bounds__(256) void k_predict_runs(PredArgs a)
{
    const int tid = threadIdx.x, sub = tid & 7;
    double st[4] = {0.0, 0.0, 0.0, 0.0};
    const int ks = a.sorted_mode, ko = 1 - ks;
    const int32_t *ids_s = a.ids + (int64_t)ks * a.n, *ids_o = a.ids + (int64_t)ko * a.n;
    const double *fs = a.fac[ks], *fo = a.fac[ko];
    const bool live = sub * 4 < a.D;                      // lanes beyond D / 4 hold zeros
    const int eoff = live ? sub * 4 : 0;
    const int64_t p0 = ((int64_t)blockIdx.x * 32 + tid / 8) * RUN;
    PairState ps[2];
    int32_t my_s[2], my_o[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        pair_load(a, p0 + 8 * q + sub, ps[q]);
        my_s[q] = ids_s[ps[q].pm]; my_o[q] = ids_o[ps[q].pm];
    }
    int32_t cur = -1;
    double4 srow = {0.0, 0.0, 0.0, 0.0};
    double keep[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (p0 + 8 * q >= a.n) break;                      // group-uniform
        int32_t is[8], io[8];
        double4 orow[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { is[u] = __shfl(my_s[q], u, 8); io[u] = __shfl(my_o[q], u, 8); }
#pragma unroll
        for (int u = 0; u < 8; u++) orow[u] = *(const double4 *)(fo + (int64_t)io[u] * a.D + eoff);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (is[u] != cur) { srow = *(const double4 *)(fs + (int64_t)is[u] * a.D + eoff); cur = is[u]; }
            double s = live ? (srow.x * orow[u].x + srow.y * orow[u].y) + (srow.z * orow[u].z + srow.w * orow[u].w) : 0.0;
            s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
            if (sub == u) keep[q] = s;
        }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) pair_finish(a, ps[q], keep[q], st);
    if (a.phase >= 0) block_stats(a, st);
}

// fixed-order sum of the per-block statistics
__global__ __launch_bounds__(256) void k_predict_final(int nblocks, const double *partial, double *stats)
{
    __shared__ double red[4][4];
    const int tid = threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = tid; b < nblocks; b += 256)
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] += partial[b * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double x = v[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
        if ((tid & 63) == 0) red[q][tid >> 6] = x;
    }
    __syncthreads();
    if (tid < 4) stats[tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

int launch_predict(bdf_ctx *ctx, PredArgs &a)
{
    if (a.n == 0) return BDF_OK;
    static const bool no_runs = getenv("BDF_PREDICT_NO_RUNS") != nullptr;       // test hook: the general kernel on sorted pairs
    if (!no_runs && a.sorted_mode >= 0 && a.n_modes == 2 && (a.D & 3) == 0 && a.D <= 32) {
        const int nblocks = (int)((a.n + 32 * RUN - 1) / (32 * RUN));
        if (a.phase >= 0) {
            void *sc;
            int rc = bdf_scratch(ctx, (size_t)nblocks * 4 * sizeof(double), &sc);
            if (rc) return rc;
            a.partial = (double *)sc;
        }
        hipLaunchKernelGGL(k_predict_runs, dim3(nblocks), dim3(256), 0, ctx->stream, a);
        if (a.phase >= 0) hipLaunchKernelGGL(k_predict_final, dim3(1), dim3(256), 0, ctx->stream, nblocks, (const double *)a.partial, a.stats);
        BDF_HIP(hipGetLastError());
        return BDF_OK;
    }
    const int64_t ntrips = (a.n + 7) / 8;
    const int nblocks = (int)std::min<int64_t>((ntrips + 31) / 32, 8192);
    if (a.phase >= 0) {
        void *sc;
        int rc = bdf_scratch(ctx, (size_t)nblocks * 4 * sizeof(double), &sc);
        if (rc) return rc;
        a.partial = (double *)sc;
    }
#define PRED(NM, VEC, NC) hipLaunchKernelGGL((k_predict<NM, VEC, NC>), dim3(nblocks), dim3(256), 0, ctx->stream, a)
#define PRED_NM(VEC, NC) do { if (a.n_modes == 2) PRED(2, VEC, NC); else if (a.n_modes == 3) PRED(3, VEC, NC); else PRED(4, VEC, NC); } while (0)
    if ((a.D & 3) != 0 || a.n_modes < 2) {
        if (a.n_modes == 1) PRED(1, 1, 1); else PRED_NM(1, 1);
    } else if (a.D <= 32) PRED_NM(4, 1);
    else PRED_NM(4, 2);
#undef PRED_NM
#undef PRED
    if (a.phase >= 0) hipLaunchKernelGGL(k_predict_final, dim3(1), dim3(256), 0, ctx->stream, nblocks, (const double *)a.partial, a.stats);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

int fill(const char *who, bdf_ctx *ctx, const bdf_pairs *p, int D, const double *const *factors, PredArgs &a)
{
    BDF_REQUIRE(ctx && p && factors, BDF_ERR_ARG, "%s: NULL argument", who);
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "%s: num_latent=%d must be in 1..%d", who, D, BDF_MAX_D);
    memset(&a, 0, sizeof(a));
    a.D = D; a.n_modes = p->n_modes; a.n = p->n; a.ids = p->ids_dev; a.values = p->values_dev;
    for (int k = 0; k < p->n_modes; k++) {
        BDF_REQUIRE(factors[k] != nullptr, BDF_ERR_ARG, "%s: factors[%d] is NULL", who, k);
        a.fac[k] = factors[k];
    }
    a.phase = -1;
    a.linear = p->baseline_dev;
    a.orig = p->orig_dev;
    a.sorted_mode = p->orig_dev ? p->sorted_mode : -1;
    return BDF_OK;
}

}  // namespace

// ---- pred_all (sampling.jl:91-97): every cell of the relation.  A workgroup takes a TILE of 16 rows of the first mode x 16 cells of
// the others (their flattened index): the first mode's rows sit in LDS, every thread owns one cell and walks the latent dimension
// in order.  (A reporting path -- predictions_full of macau.jl:145-147 -- not the sweep: plain fp64, no matrix cores.)
namespace {
struct PredAllArgs {
    int D, n_modes;
    int64_t dims[BDF_MAX_MODES];
    const double *fac[BDF_MAX_MODES];
    double mean;
    double *out;
    int64_t rest;                  // cells of the modes behind the first: dims[1] * ... * dims[n - 1]
};
__global__ __launch_bounds__(256) void k_predict_all(PredAllArgs a)
{
    __shared__ double u[16][BDF_MAX_D + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * 16, c = (int64_t)blockIdx.x * 16 + tx;
    for (int e = threadIdx.x; e < 16 * a.D; e += 256) {
        const int r = e / a.D, d = e - r * a.D;
        u[r][d] = i0 + r < a.dims[0] ? a.fac[0][(i0 + r) * a.D + d] : 0.0;
    }
    __syncthreads();
    if (c >= a.rest || i0 + ty >= a.dims[0]) return;
    // the cell's indices in the modes behind the first (the last one fastest)
    const double *row[BDF_MAX_MODES];
    int64_t q = c;
#pragma unroll
    for (int k = BDF_MAX_MODES - 1; k >= 1; k--)
        if (k < a.n_modes) { row[k] = a.fac[k] + (q % a.dims[k]) * a.D; q /= a.dims[k]; }
    double s = 0.0;
    for (int d = 0; d < a.D; d++) {
        double pr = u[ty][d];
#pragma unroll
        for (int k = 1; k < BDF_MAX_MODES; k++)
            if (k < a.n_modes) pr *= row[k][d];
        s += pr;
    }
    a.out[(i0 + ty) * a.rest + c] = s + a.mean;
}
}  // namespace

extern "C" int bdf_predict_all(bdf_ctx *ctx, int n_modes, const int64_t *dims, int D, const double *const *factors,
                               double mean_value, double *out)
{
    BDF_REQUIRE(ctx && dims && factors && out, BDF_ERR_ARG, "bdf_predict_all: NULL argument");
    BDF_REQUIRE(n_modes >= 2 && n_modes <= BDF_MAX_MODES, BDF_ERR_ARG, "bdf_predict_all: n_modes=%d must be in 2..%d", n_modes, BDF_MAX_MODES);
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_predict_all: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    PredAllArgs a;
    memset(&a, 0, sizeof(a));
    a.D = D; a.n_modes = n_modes; a.mean = mean_value; a.out = out; a.rest = 1;
    for (int k = 0; k < n_modes; k++) {
        BDF_REQUIRE(dims[k] >= 0 && factors[k] != nullptr, BDF_ERR_ARG, "bdf_predict_all: dims[%d] < 0 or factors[%d] NULL", k, k);
        a.dims[k] = dims[k]; a.fac[k] = factors[k];
        if (k >= 1) a.rest *= dims[k];
    }
    if (a.dims[0] == 0 || a.rest == 0) return BDF_OK;
    const int64_t gx = (a.rest + 15) / 16, gy = (a.dims[0] + 15) / 16;
    BDF_REQUIRE(gx < (int64_t)0x7fffffff && gy <= 65535 * (int64_t)1024, BDF_ERR_BOUNDS, "bdf_predict_all: %lld x %lld cells are too many for one launch",
                (long long)a.dims[0], (long long)a.rest);
    BDF_HIP(hipSetDevice(ctx->device));
    // (grid.y is limited to 65,535 blocks: the first mode in slabs of that many tiles)
    for (int64_t y0 = 0; y0 < gy; y0 += 65535) {
        PredAllArgs b = a;
        const int64_t ny = std::min<int64_t>(65535, gy - y0);
        b.fac[0] = a.fac[0] + y0 * 16 * D;
        b.dims[0] = std::min<int64_t>(a.dims[0] - y0 * 16, ny * 16);
        b.out = a.out + y0 * 16 * a.rest;
        hipLaunchKernelGGL(k_predict_all, dim3((unsigned)gx, (unsigned)ny), dim3(256), 0, ctx->stream, b);
    }
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

extern "C" int bdf_pairs_create(bdf_ctx *ctx, int n_modes, int64_t n, const void *ids, int id_bytes,
                                const double *values, bdf_pairs **out)
{
    BDF_REQUIRE(ctx && out, BDF_ERR_ARG, "bdf_pairs_create: NULL argument");
    BDF_REQUIRE(n_modes >= 2 && n_modes <= BDF_MAX_MODES, BDF_ERR_ARG, "bdf_pairs_create: n_modes=%d must be in 2..%d", n_modes, BDF_MAX_MODES);
    BDF_REQUIRE(id_bytes == 4 || id_bytes == 8, BDF_ERR_ARG, "bdf_pairs_create: id_bytes must be 4 or 8");
    BDF_REQUIRE(n >= 0 && (n == 0 || (ids && values)), BDF_ERR_ARG, "bdf_pairs_create: ids/values NULL");
    BDF_HIP(hipSetDevice(ctx->device));
    std::vector<int32_t> h((size_t)n * n_modes);
    for (size_t q = 0; q < h.size(); q++) {
        int64_t v = id_bytes == 8 ? ((const int64_t *)ids)[q] : (int64_t)((const int32_t *)ids)[q];
        BDF_REQUIRE(v >= 1 && v < (int64_t)0x7fffffff, BDF_ERR_BOUNDS, "bdf_pairs_create: id %lld out of range", (long long)v);
        h[q] = (int32_t)(v - 1);
    }
    bdf_pairs *p = new bdf_pairs();
    p->ctx = ctx; p->n_modes = n_modes; p->n = n; p->count = 0.0; p->baseline_dev = nullptr; p->orig_dev = nullptr; p->sorted_mode = -1;
    p->ids_dev = nullptr; p->values_dev = nullptr; p->avg_dev = nullptr; p->sq_dev = nullptr;
    struct Guard { bdf_pairs *p; ~Guard() { if (p) bdf_pairs_destroy(p); } } guard{p};        // error paths free what was allocated
    p->ids_host = h;
    p->values_host.assign(values, values + (n ? n : 0));
    size_t nb = std::max<size_t>((size_t)n * sizeof(double), 8);
    BDF_HIP(hipMalloc((void **)&p->ids_dev, std::max<size_t>(h.size() * sizeof(int32_t), 8)));
    BDF_HIP(hipMalloc((void **)&p->values_dev, nb));
    BDF_HIP(hipMalloc((void **)&p->avg_dev, nb));
    BDF_HIP(hipMalloc((void **)&p->sq_dev, nb));
    if (n) {
        BDF_HIP(hipMemcpy(p->ids_dev, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        BDF_HIP(hipMemcpy(p->values_dev, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    }
    // (on the context's stream and waited for: the pairs may be updated on another stream next, and a plain hipMemset may still
    // be pending on the NULL stream when it returns)
    BDF_HIP(hipMemsetAsync(p->avg_dev, 0, nb, ctx->stream));
    BDF_HIP(hipMemsetAsync(p->sq_dev, 0, nb, ctx->stream));
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    guard.p = nullptr;
    *out = p;
    return BDF_OK;
}

extern "C" int bdf_pairs_destroy(bdf_pairs *p)
{
    if (!p) return BDF_OK;
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    hipFree(p->ids_dev); hipFree(p->values_dev); hipFree(p->avg_dev); hipFree(p->sq_dev);
    if (p->orig_dev) hipFree(p->orig_dev);
    delete p;
    return BDF_OK;
}

extern "C" int bdf_predict(bdf_ctx *ctx, const bdf_pairs *p, int D, const double *const *factors,
                           double mean_value, double *out)
{
    PredArgs a;
    int rc = fill("bdf_predict", ctx, p, D, factors, a);
    if (rc) return rc;
    BDF_REQUIRE(out != nullptr, BDF_ERR_ARG, "bdf_predict: out is NULL");
    a.mean = mean_value; a.out = out;
    return launch_predict(ctx, a);
}

// udot + mean_value, whatever baseline the pairs carry (sample_beta_rel needs the residual against the plain mean)
int bdf_predict_plain(bdf_ctx *ctx, const bdf_pairs *p, int D, const double *const *factors, double mean_value, double *out)
{
    PredArgs a;
    int rc = fill("bdf_predict", ctx, p, D, factors, a);
    if (rc) return rc;
    a.mean = mean_value; a.linear = nullptr; a.out = out;
    return launch_predict(ctx, a);
}

extern "C" int bdf_predict_update(bdf_ctx *ctx, bdf_pairs *p, int D, const double *const *factors,
                                  double mean_value, int phase, double clamp_lo, double clamp_hi,
                                  double class_cut, double *stats_out)
{
    PredArgs a;
    int rc = fill("bdf_predict_update", ctx, p, D, factors, a);
    if (rc) return rc;
    BDF_REQUIRE(stats_out != nullptr, BDF_ERR_ARG, "bdf_predict_update: stats_out is NULL");
    BDF_REQUIRE(phase >= 0 && phase <= 2, BDF_ERR_ARG, "bdf_predict_update: phase must be 0, 1 or 2");
    a.mean = mean_value; a.avg = p->avg_dev; a.sq = p->sq_dev; a.phase = phase; a.count = p->count;
    a.clamp_lo = clamp_lo; a.clamp_hi = clamp_hi; a.cut = class_cut; a.stats = stats_out;
    rc = launch_predict(ctx, a);
    if (rc) return rc;
    if (phase == 1) p->count = 1.0;
    else if (phase == 2) p->count += 1.0;
    return BDF_OK;
}

extern "C" int bdf_predict_sse(bdf_ctx *ctx, const bdf_pairs *p, int D, const double *const *factors, double mean_value,
                               const double *linear_values, double *stats_out)
{
    // sum over the pairs of (value - pred)^2 with pred = udot + (linear_values[pair] | mean_value): the err' err of
    // sample_alpha (macau.jl:86-87); stats_out as bdf_predict_update's, [1] is the sum of squares
    PredArgs a;
    int rc = fill("bdf_predict_sse", ctx, p, D, factors, a);
    if (rc) return rc;
    BDF_REQUIRE(stats_out != nullptr, BDF_ERR_ARG, "bdf_predict_sse: stats_out is NULL");
    a.mean = mean_value; if (linear_values) a.linear = linear_values; a.phase = 3; a.count = 0.0;
    a.clamp_lo = 1.0; a.clamp_hi = 0.0; a.cut = 0.0; a.stats = stats_out;
    return launch_predict(ctx, a);
}

extern "C" int bdf_pairs_sort(bdf_pairs *p, int mode)
{
    // Store the pairs sorted by their id in `mode` (stable): consecutive pairs then share that mode's factor row, which
    // the 8 lanes of the next pair find in cache -- half the gather traffic of a prediction update.  The caller's order
    // is kept in every interface: bdf_predict's out and the baseline are indexed through the permutation, bdf_pairs_order
    // returns it for the running state (bdf_pairs_state stays in storage order).  Only before the first update.
    BDF_REQUIRE(p != nullptr, BDF_ERR_ARG, "bdf_pairs_sort: NULL argument");
    BDF_REQUIRE(mode >= 0 && mode < p->n_modes, BDF_ERR_ARG, "bdf_pairs_sort: mode %d out of range", mode);
    BDF_REQUIRE(p->count == 0.0 && p->orig_dev == nullptr, BDF_ERR_ARG, "bdf_pairs_sort: the pairs already hold prediction state or are sorted");
    const int64_t n = p->n;
    if (n == 0) return BDF_OK;
    BDF_HIP(hipSetDevice(p->ctx->device));
    std::vector<int32_t> perm((size_t)n);
    for (int64_t i = 0; i < n; i++) perm[(size_t)i] = (int32_t)i;
    const int32_t *key = p->ids_host.data() + (size_t)mode * n;
    std::stable_sort(perm.begin(), perm.end(), [key](int32_t x, int32_t y) { return key[x] < key[y]; });
    std::vector<int32_t> ids((size_t)n * p->n_modes);
    std::vector<double> vals((size_t)n);
    for (int k = 0; k < p->n_modes; k++)
        for (int64_t i = 0; i < n; i++) ids[(size_t)k * n + i] = p->ids_host[(size_t)k * n + perm[(size_t)i]];
    for (int64_t i = 0; i < n; i++) vals[(size_t)i] = p->values_host[(size_t)perm[(size_t)i]];
    BDF_HIP(hipStreamSynchronize(p->ctx->stream));
    BDF_HIP(hipMemcpy(p->ids_dev, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    BDF_HIP(hipMemcpy(p->values_dev, vals.data(), vals.size() * sizeof(double), hipMemcpyHostToDevice));
    BDF_HIP(hipMalloc((void **)&p->orig_dev, (size_t)n * sizeof(int32_t)));
    BDF_HIP(hipMemcpy(p->orig_dev, perm.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    p->orig_host = perm;
    p->sorted_mode = mode;
    return BDF_OK;
}

extern "C" int bdf_pairs_order(const bdf_pairs *p, int64_t *orig_host)
{
    // orig_host[i] = the caller's index of the pair stored at position i (identity unless bdf_pairs_sort was called)
    BDF_REQUIRE(p && orig_host, BDF_ERR_ARG, "bdf_pairs_order: NULL argument");
    for (int64_t i = 0; i < p->n; i++) orig_host[i] = p->orig_host.empty() ? i : (int64_t)p->orig_host[(size_t)i];
    return BDF_OK;
}

extern "C" int bdf_pairs_set_baseline(bdf_pairs *p, const double *baseline)
{
    BDF_REQUIRE(p != nullptr, BDF_ERR_ARG, "bdf_pairs_set_baseline: NULL argument");
    p->baseline_dev = baseline;
    return BDF_OK;
}

extern "C" int bdf_pairs_state(const bdf_pairs *p, double **avg, double **sq, int64_t *n)
{
    BDF_REQUIRE(p && avg && sq && n, BDF_ERR_ARG, "bdf_pairs_state: NULL argument");
    *avg = p->avg_dev; *sq = p->sq_dev; *n = p->n;
    return BDF_OK;
}
