// bdf_api.hip -- C-ABI entry points: context, device memory, IndexedDF -> device CSR, row sampling front-end
#include <chrono>
#include "bdf_common.h"
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <numeric>
#include <atomic>

static thread_local char g_err[512] = "";

void bdf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *bdf_last_error(void) { return g_err; }
extern "C" int bdf_version(void) { return 100; }

// ---- context -----------------------------------------------------------------------------
extern "C" int bdf_ctx_create(int device, void *stream, uint64_t seed, bdf_ctx **out)
{
    BDF_REQUIRE(out != nullptr, BDF_ERR_ARG, "bdf_ctx_create: out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        bdf_set_error("bdf_ctx_create: no HIP device visible (this library has no CPU path)");
        return BDF_ERR_NOGPU;
    }
    BDF_REQUIRE(device >= 0 && device < ndev, BDF_ERR_ARG, "bdf_ctx_create: device %d out of range (0..%d)", device, ndev - 1);
    BDF_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    BDF_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        bdf_set_error("bdf_ctx_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return BDF_ERR_NOGPU;
    }
    bdf_ctx *c = new bdf_ctx();
    c->sweep_dev = nullptr; c->flag_dev = nullptr; c->flag_host = nullptr; c->scratch = nullptr; c->scratch2 = nullptr; c->cg_status = nullptr; c->cg_part = nullptr;
    c->hyper_fuse = false; c->hyper_partial = nullptr; c->hyper_nblocks = 0; c->hyper_sumU = c->hyper_UUt = nullptr;
    c->hyper_chain = false; c->hyper_count = nullptr; c->hyper_chain_draws = nullptr;
    c->own_stream = false; c->stream = nullptr;
    struct Guard { bdf_ctx *c; ~Guard() { if (c) bdf_ctx_destroy(c); } } guard{c};        // error paths free what was allocated
    c->device = device;
    c->seed = seed;
    c->own_stream = false;
    c->stream = (hipStream_t)stream;          // NULL is the device's default stream
    BDF_HIP(hipMalloc((void **)&c->sweep_dev, sizeof(uint32_t)));
    BDF_HIP(hipHostMalloc((void **)&c->flag_host, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->flag_host, 0, 16 * sizeof(int));
    BDF_HIP(hipHostGetDevicePointer((void **)&c->flag_dev, c->flag_host, 0));
    BDF_HIP(hipMemsetAsync(c->sweep_dev, 0, sizeof(uint32_t), c->stream));
    c->time_start = c->time_stop = nullptr;
    c->time_h_start = c->time_h_stop = nullptr;
    c->sweep_host = 0;
    c->rows_span = nullptr;
    c->rows_ready = nullptr; c->rows_ready_want = 0; c->hyper_ready = nullptr; c->hyper_ready_value = 0;
    c->rows_done = nullptr; c->rows_done_added = -1; c->hyper_wait = nullptr; c->hyper_wait_target = 0;
    c->warnings = 0;
    c->reserve_cus = 0;
    c->on_reserved = 0;
    c->skip_flag = nullptr;
    c->cg_status = nullptr; c->cg_part = nullptr;
    c->hyper_fuse = false; c->hyper_partial = nullptr; c->hyper_nblocks = 0; c->hyper_sumU = c->hyper_UUt = nullptr;
    c->hyper_chain = false; c->hyper_count = nullptr; c->hyper_chain_draws = nullptr;
    c->cg_gen = 0;
    c->cg_bar = nullptr;
    c->rows_dispatch = nullptr;
    c->scratch = nullptr;
    c->scratch2 = nullptr;
    c->scratch2_bytes = 0;
    c->scratch_bytes = 0;
    c->item_size = 192;
    c->piece_size = 128;
    c->item_auto = !getenv("BDF_ITEM_FIXED");
    c->small_max = getenv("BDF_K1_SMALL") ? atoi(getenv("BDF_K1_SMALL")) : 48;
    c->small_min_rows = getenv("BDF_K1_SMALL_MIN_ROWS") ? atoll(getenv("BDF_K1_SMALL_MIN_ROWS")) : 8192;
    c->lr_max = getenv("BDF_LOWRANK") ? atoi(getenv("BDF_LOWRANK")) : -1;
    c->lr_min_rows = getenv("BDF_LOWRANK_MIN_ROWS") ? atoll(getenv("BDF_LOWRANK_MIN_ROWS")) : 8192;
    // K1c (k_rows_col.hip): on by default with pieces of at most 128 observations; BDF_K1_COL=0: off, BDF_K1_COL=<n>: that piece size
    c->col_piece = getenv("BDF_K1_COL") ? atoi(getenv("BDF_K1_COL")) : 128;
    if (c->col_piece == 1) c->col_piece = 128;
    if (c->col_piece != 0) c->col_piece = std::min(4096, std::max(8, c->col_piece));
    c->col_explicit = false;
    c->lr_T = nullptr; c->lr_vt = nullptr; c->lr_vt_bytes = 0; c->lr_mrows = nullptr; c->lr_mrows_bytes = 0;
    c->lr_key_fac = c->lr_key_Lambda = c->lr_key_mu = nullptr; c->lr_key_sweep = c->lr_key_tag = 0; c->lr_key_D = 0; c->lr_key_M = 0;
    {
        const char *force = getenv("BDF_GATHER");
        c->gather_mode = force && !strcmp(force, "general") ? 1 : (force && !strcmp(force, "wide") ? 2 : 0);
    }
    guard.c = nullptr;
    *out = c;
    return BDF_OK;
}

extern "C" int bdf_ctx_destroy(bdf_ctx *ctx)
{
    if (!ctx) return BDF_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    bdf_plans_release(ctx, 0);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->sweep_dev) hipFree(ctx->sweep_dev);
    if (ctx->flag_host) hipHostFree(ctx->flag_host);
    if (ctx->scratch2) hipFree(ctx->scratch2);
    if (ctx->cg_status) hipHostFree((void *)ctx->cg_status);
    if (ctx->cg_part) hipFree(ctx->cg_part);
    if (ctx->cg_bar) hipFree(ctx->cg_bar);
    delete ctx->rows_dispatch;
    if (ctx->hyper_count) hipFree(ctx->hyper_count);
    if (ctx->lr_T) hipFree(ctx->lr_T);
    if (ctx->lr_vt) hipFree(ctx->lr_vt);
    if (ctx->lr_mrows) hipFree(ctx->lr_mrows);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return BDF_OK;
}

int bdf_scratch(bdf_ctx *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->scratch_bytes) {
        // growing frees the old block: wait for work that may still read it
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) BDF_HIP(hipFree(ctx->scratch));
        size_t nb = std::max(bytes, ctx->scratch_bytes * 2);
        nb = (nb + 255) & ~(size_t)255;
        BDF_HIP(hipMalloc(&ctx->scratch, nb));
        ctx->scratch_bytes = nb;
    }
    *out = ctx->scratch;
    return BDF_OK;
}

int bdf_scratch2(bdf_ctx *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->scratch2_bytes) {
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch2) BDF_HIP(hipFree(ctx->scratch2));
        size_t nb = std::max(bytes, ctx->scratch2_bytes * 2);
        nb = (nb + 255) & ~(size_t)255;
        BDF_HIP(hipMalloc(&ctx->scratch2, nb));
        ctx->scratch2_bytes = nb;
    }
    *out = ctx->scratch2;
    return BDF_OK;
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }
__global__ void k_inc_u32(uint32_t *p) { *p = *p + 1; }

extern "C" int bdf_ctx_set_sweep(bdf_ctx *ctx, uint32_t sweep)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_ctx_set_sweep: ctx is NULL");
    ctx->sweep_host = sweep;          // passed to every launch by value (no launch, no device round trip)
    return BDF_OK;
}

extern "C" int bdf_ctx_advance_sweep(bdf_ctx *ctx)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_ctx_advance_sweep: ctx is NULL");
    ctx->sweep_host++;
    return BDF_OK;
}

extern "C" int bdf_ctx_set_item_size(bdf_ctx *ctx, int observations)
{
    BDF_REQUIRE(ctx && (observations == 0 || (observations >= 8 && observations <= (1 << 20))), BDF_ERR_ARG, "bdf_ctx_set_item_size: 0 (automatic) or 8..2^20 observations");
    if (observations == 0) {          // the default: 192 / 128, larger for launches with hundreds of waves per resident slot
        ctx->item_size = 192; ctx->piece_size = 128; ctx->item_auto = !getenv("BDF_ITEM_FIXED");
        return BDF_OK;
    }
    ctx->item_size = observations;
    ctx->piece_size = std::max(8, observations * 2 / 3);
    ctx->item_auto = false;
    return BDF_OK;
}

extern "C" int bdf_ctx_set_gather(bdf_ctx *ctx, int mode)
{
    BDF_REQUIRE(ctx && mode >= 0 && mode <= 2, BDF_ERR_ARG, "bdf_ctx_set_gather: mode must be 0 (auto), 1 (general) or 2 (64-bit offsets)");
    ctx->gather_mode = mode;
    return BDF_OK;
}

extern "C" int bdf_ctx_set_small_rows(bdf_ctx *ctx, int max_observations, int64_t min_rows)
{
    BDF_REQUIRE(ctx && max_observations >= 0 && min_rows >= 0, BDF_ERR_ARG, "bdf_ctx_set_small_rows: bad argument");
    ctx->small_max = max_observations;
    ctx->small_min_rows = min_rows;
    return BDF_OK;
}

extern "C" int bdf_ctx_set_lowrank(bdf_ctx *ctx, int max_observations, int64_t min_rows)
{
    BDF_REQUIRE(ctx && max_observations >= -1 && max_observations <= 32 && min_rows >= 0, BDF_ERR_ARG,
                "bdf_ctx_set_lowrank: max_observations must be -1 (default), 0 (off) or 1..32");
    ctx->lr_max = max_observations;
    ctx->lr_min_rows = min_rows;
    return BDF_OK;
}

// slot (dev, 64 x 2 uint64): every pair set to {~0, 0} by the caller; the next K1c launch of this context leaves {earliest start,
// latest end} of its waves w = s mod 64 in pair s, in ticks of the 100 MHz clock the XCDs share (s_memrealtime): min / max over the
// pairs = the launch's duration without events around it
extern "C" int bdf_ctx_span_next_rows(bdf_ctx *ctx, void *slot_dev)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_ctx_span_next_rows: NULL context");
    ctx->rows_span = (unsigned long long *)slot_dev;
    return BDF_OK;
}

extern "C" int bdf_ctx_rows_dispatch(const bdf_ctx *ctx, uint32_t entity_tag, int64_t out[6])
{
    BDF_REQUIRE(ctx && out, BDF_ERR_ARG, "bdf_ctx_rows_dispatch: NULL argument");
    BDF_REQUIRE(ctx->rows_dispatch && ctx->rows_dispatch->count(entity_tag), BDF_ERR_ARG,
                "bdf_ctx_rows_dispatch: no row launch under entity_tag %u on this context", entity_tag);
    const std::array<int64_t, 7> &r = ctx->rows_dispatch->at(entity_tag);
    for (int k = 0; k < 6; k++) out[k] = r[(size_t)k + 1];
    return BDF_OK;
}

extern "C" int bdf_ctx_set_col_rows(bdf_ctx *ctx, int max_piece)
{
    BDF_REQUIRE(ctx && (max_piece == 0 || max_piece == -1 || (max_piece >= 8 && max_piece <= 4096)), BDF_ERR_ARG,
                "bdf_ctx_set_col_rows: -1 (default), 0 (off) or 8..4096 observations per piece");
    if (max_piece == -1) {            // the default again: BDF_K1_COL or 128, for launches whose item size the caller has not set
        ctx->col_piece = getenv("BDF_K1_COL") ? atoi(getenv("BDF_K1_COL")) : 128;
        if (ctx->col_piece == 1) ctx->col_piece = 128;
        if (ctx->col_piece != 0) ctx->col_piece = std::min(4096, std::max(8, ctx->col_piece));
        ctx->col_explicit = false;
        return BDF_OK;
    }
    ctx->col_piece = max_piece;
    ctx->col_explicit = true;
    return BDF_OK;
}

extern "C" int bdf_ctx_set_piece_size(bdf_ctx *ctx, int observations)
{
    BDF_REQUIRE(ctx && observations >= 8 && observations <= ctx->item_size, BDF_ERR_ARG,
                "bdf_ctx_set_piece_size: 8..item size (%d) observations", ctx ? ctx->item_size : 0);
    ctx->piece_size = observations;
    ctx->item_auto = false;
    return BDF_OK;
}

extern "C" int bdf_ctx_sync(bdf_ctx *ctx)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_ctx_sync: ctx is NULL");
    // a short spin before the blocking wait (a sleeping thread wakes 10-40 us after the stream has drained)
    hipError_t st = hipErrorNotReady;
    const auto t_spin = std::chrono::steady_clock::now();
    while ((st = hipStreamQuery(ctx->stream)) == hipErrorNotReady &&
           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_spin).count() < 300.0) { }
    if (st == hipErrorNotReady) st = hipStreamSynchronize(ctx->stream);
    BDF_HIP(st);
    int flag = __atomic_load_n(ctx->flag_host, __ATOMIC_ACQUIRE);
    if (flag) {
        __atomic_store_n(ctx->flag_host, 0, __ATOMIC_RELEASE);      // (the stream is idle: nothing of this context is in flight)
        ctx->warnings |= (uint32_t)flag & BDF_WARN_CG_MAXITER;
        flag &= ~(int)BDF_WARN_CG_MAXITER;
    }
    if (flag) {
        if (flag & 16) {
            bdf_set_error("row sampler: a split row's pieces did not all arrive in time (flag %d)", flag);
            return BDF_ERR_HIP;
        }
        // bits: 1 a row's P_i, 2 the Normal-Wishart draw, 4 Lambda of the beta noise, 8 FF + lambda I of the direct solve
        bdf_set_error("a matrix that must be positive definite was not (flag %d: %s%s%s%s)", flag, flag & 1 ? "row system " : "",
                      flag & 2 ? "hyperprior " : "", flag & 4 ? "noise precision " : "", flag & 8 ? "FF + lambda I" : "");
        return BDF_ERR_NOTPD;
    }
    return BDF_OK;
}

extern "C" int bdf_ctx_warnings(bdf_ctx *ctx, uint32_t *bits_out)
{
    BDF_REQUIRE(ctx && bits_out, BDF_ERR_ARG, "bdf_ctx_warnings: NULL argument");
    *bits_out = ctx->warnings;
    ctx->warnings = 0;
    return BDF_OK;
}

// ---- timing of a row-kernel launch by HIP events carried by the kernel's own dispatch packet ------------------------
extern "C" int bdf_event_create(void **ev)
{
    BDF_REQUIRE(ev, BDF_ERR_ARG, "bdf_event_create: NULL argument");
    hipEvent_t e;
    BDF_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return BDF_OK;
}

extern "C" int bdf_event_destroy(void *ev)
{
    if (ev) BDF_HIP(hipEventDestroy((hipEvent_t)ev));
    return BDF_OK;
}

extern "C" int bdf_event_elapsed_us(void *start, void *stop, double *us)
{
    BDF_REQUIRE(start && stop && us, BDF_ERR_ARG, "bdf_event_elapsed_us: NULL argument");
    BDF_HIP(hipEventSynchronize((hipEvent_t)stop));
    float ms = 0.f;
    BDF_HIP(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
    *us = 1e3 * (double)ms;
    return BDF_OK;
}

extern "C" int bdf_ctx_time_next_rows(bdf_ctx *ctx, void *start, void *stop)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_ctx_time_next_rows: ctx is NULL");
    ctx->time_start = (hipEvent_t)start;
    ctx->time_stop = (hipEvent_t)stop;
    return BDF_OK;
}

extern "C" int bdf_ctx_time_next_hyper(bdf_ctx *ctx, void *start, void *stop)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_ctx_time_next_hyper: ctx is NULL");
    ctx->time_h_start = (hipEvent_t)start;
    ctx->time_h_stop = (hipEvent_t)stop;
    return BDF_OK;
}

extern "C" int bdf_dev_alloc(bdf_ctx *ctx, size_t bytes, void **dptr)
{
    BDF_REQUIRE(ctx && dptr, BDF_ERR_ARG, "bdf_dev_alloc: NULL argument");
    BDF_HIP(hipSetDevice(ctx->device));
    BDF_HIP(hipMalloc(dptr, bytes ? bytes : 8));
    return BDF_OK;
}

extern "C" int bdf_dev_free(bdf_ctx *ctx, void *dptr)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_dev_free: ctx is NULL");
    if (dptr) {
        BDF_HIP(hipStreamSynchronize(ctx->stream));
        BDF_HIP(hipFree(dptr));
    }
    return BDF_OK;
}

extern "C" int bdf_h2d(bdf_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    BDF_REQUIRE(ctx && (bytes == 0 || (dst && src)), BDF_ERR_ARG, "bdf_h2d: NULL argument");
    if (bytes) {
        BDF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        BDF_HIP(hipStreamSynchronize(ctx->stream));   // src is caller-owned for the call only
    }
    return BDF_OK;
}

extern "C" int bdf_d2h(bdf_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    BDF_REQUIRE(ctx && (bytes == 0 || (dst && src)), BDF_ERR_ARG, "bdf_d2h: NULL argument");
    if (bytes) {
        BDF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        BDF_HIP(hipStreamSynchronize(ctx->stream));
    }
    return BDF_OK;
}

template <typename T>
static int upload(bdf_ctx *ctx, const std::vector<T> &v, T **dptr)
{
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 8);
    BDF_HIP(hipMalloc((void **)dptr, bytes));
    if (!v.empty()) BDF_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return BDF_OK;
}

// ---- a1: IndexedDF / FastIDF ------------------------------------------------------------
// IndexedDF.jl:10-19 pushes COO row i onto index[mode][id] in row order: a stable counting sort.
static inline int64_t id_at(const void *ids, int id_bytes, int64_t nnz, int64_t i, int m)
{
    size_t off = (size_t)m * (size_t)nnz + (size_t)i;
    return id_bytes == 8 ? ((const int64_t *)ids)[off] : (int64_t)((const int32_t *)ids)[off];
}

extern "C" int bdf_index_build(int n_modes, const int64_t *dims, int64_t nnz, const void *ids, int id_bytes,
                               int64_t *const *rowptr, int64_t *const *rowids)
{
    BDF_REQUIRE(dims && rowptr && rowids, BDF_ERR_ARG, "bdf_index_build: NULL argument");
    BDF_REQUIRE(n_modes >= 1 && n_modes <= BDF_MAX_MODES, BDF_ERR_ARG, "bdf_index_build: n_modes=%d must be in 1..%d", n_modes, BDF_MAX_MODES);
    BDF_REQUIRE(id_bytes == 4 || id_bytes == 8, BDF_ERR_ARG, "bdf_index_build: id_bytes must be 4 or 8");
    BDF_REQUIRE(nnz >= 0 && (nnz == 0 || ids), BDF_ERR_ARG, "bdf_index_build: ids NULL");
    for (int m = 0; m < n_modes; m++) {
        const int64_t N = dims[m];
        BDF_REQUIRE(N >= 0 && rowptr[m] && (nnz == 0 || rowids[m]), BDF_ERR_ARG, "bdf_index_build: bad mode %d", m);
        int64_t *rp = rowptr[m], *ri = rowids[m];
        for (int64_t j = 0; j <= N; j++) rp[j] = 0;
        for (int64_t i = 0; i < nnz; i++) {
            int64_t j = id_at(ids, id_bytes, nnz, i, m);
            if (j < 1 || j > N) {
                bdf_set_error("id %lld of mode %d at row %lld outside 1..%lld (BoundsError)", (long long)j, m + 1,
                              (long long)(i + 1), (long long)N);
                return BDF_ERR_BOUNDS;
            }
            rp[j]++;
        }
        for (int64_t j = 0; j < N; j++) rp[j + 1] += rp[j];
        std::vector<int64_t> cursor(rp, rp + N);
        for (int64_t i = 0; i < nnz; i++) ri[cursor[(size_t)id_at(ids, id_bytes, nnz, i, m) - 1]++] = i + 1;
    }
    return BDF_OK;
}

// ---- row layout of an entity shared by several GPUs (host-only) ------------------------------------------------------------
// Rows in falling order of degree (stable) are dealt to the ranks round-robin (the reference deals rows i:P:N to its P
// workers, src/sampling.jl:154), a rank's rows round-robin to `chunks` chunks; row r of chunk c of rank p sits at internal
// position (c * world + p) * cmax + r: every chunk is one contiguous, rank-major region of the factor matrix -- what an
// in-place all-gather fills -- and the row kernel of chunk c + 1 can run while chunk c is exchanged.
extern "C" int bdf_layout_build(int64_t N, const int64_t *degree, int world, int chunks, int32_t *pos_out, int64_t *cmax_out)
{
    BDF_REQUIRE(N >= 0 && N < (int64_t)0x7fffffff && (N == 0 || (degree && pos_out)) && cmax_out, BDF_ERR_ARG, "bdf_layout_build: bad argument");
    BDF_REQUIRE(world >= 1 && chunks >= 1, BDF_ERR_ARG, "bdf_layout_build: world and chunks must be positive");
    const int64_t per_rank = (N + world - 1) / world;
    const int64_t cmax = std::max<int64_t>(1, (per_rank + chunks - 1) / chunks);
    BDF_REQUIRE(cmax * world * chunks < (int64_t)0x7fffffff, BDF_ERR_ARG, "bdf_layout_build: layout too large");
    std::vector<int32_t> ord((size_t)N);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return degree[x] > degree[y]; });
    for (int64_t s = 0; s < N; s++) {
        const int64_t p = s % world, r = s / world, c = r % chunks, i = r / chunks;
        pos_out[ord[(size_t)s]] = (int32_t)((c * world + p) * cmax + i);
    }
    *cmax_out = cmax;
    return BDF_OK;
}

static int relation_create(bdf_ctx *ctx, int n_modes, const int64_t *dims, int64_t nnz, const void *ids, int id_bytes,
                           const double *values, const int32_t *const *pos, const int64_t *cmax, int rank, int world, int chunks,
                           bdf_rel **out)
{
    BDF_REQUIRE(ctx && dims && out, BDF_ERR_ARG, "bdf_relation_create: NULL argument");
    BDF_REQUIRE(n_modes >= 2 && n_modes <= BDF_MAX_MODES, BDF_ERR_ARG,
                "bdf_relation_create: n_modes=%d must be in 2..%d", n_modes, BDF_MAX_MODES);
    BDF_REQUIRE(id_bytes == 4 || id_bytes == 8, BDF_ERR_ARG, "bdf_relation_create: id_bytes must be 4 or 8");
    BDF_REQUIRE(nnz >= 0 && nnz < (int64_t)0x7fffffff, BDF_ERR_ARG, "bdf_relation_create: nnz=%lld unsupported", (long long)nnz);
    BDF_REQUIRE(nnz == 0 || (ids && values), BDF_ERR_ARG, "bdf_relation_create: ids/values NULL");
    for (int m = 0; m < n_modes; m++)
        BDF_REQUIRE(dims[m] >= 0 && dims[m] < (int64_t)0x7fffffff, BDF_ERR_ARG, "bdf_relation_create: dims[%d]=%lld unsupported", m, (long long)dims[m]);
    const bool sharded = pos != nullptr;
    if (sharded) {
        BDF_REQUIRE(cmax && world >= 1 && rank >= 0 && rank < world && chunks >= 1, BDF_ERR_ARG, "bdf_relation_create_sharded: bad rank / world / chunks");
        for (int m = 0; m < n_modes; m++) BDF_REQUIRE(pos[m] != nullptr && cmax[m] >= 1, BDF_ERR_ARG, "bdf_relation_create_sharded: mode %d has no layout", m);
    }
    BDF_HIP(hipSetDevice(ctx->device));
    auto idat = [&](int64_t i, int m) -> int64_t { return id_at(ids, id_bytes, nnz, i, m); };

    static std::atomic<uint64_t> next_serial{1};
    bdf_rel *r = new bdf_rel();
    struct Guard { bdf_rel *r; ~Guard() { if (r) bdf_relation_destroy(r); } } guard{r};        // error paths free what was built
    r->ctx = ctx;
    r->serial = next_serial.fetch_add(1);
    r->n_modes = n_modes;
    r->nnz = nnz;
    r->sharded = sharded ? 1 : 0; r->rank = sharded ? rank : 0; r->world = sharded ? world : 1; r->chunks = sharded ? chunks : 1;
    for (int m = 0; m < n_modes; m++) {
        r->dims[m] = dims[m];
        r->nint[m] = sharded ? cmax[m] * world * chunks : dims[m];
        r->idx[m].rowptr_dev = nullptr; r->idx[m].colidx_dev = nullptr; r->idx[m].vals_dev = nullptr; r->idx[m].perm_dev = nullptr;
        r->idx[m].order_dev = nullptr; r->idx[m].own_nnz = 0; r->idx[m].packed_dev = nullptr;
    }
    r->n_codes = 0; r->table_dev = nullptr;
    double s = 0.0;
    for (int64_t i = 0; i < nnz; i++) s += values[i];
    r->value_mean = nnz ? s / (double)nnz : NAN;
    // the distinct values, if there are at most 256 of them (two-mode relations: K1's coded variant)
    std::vector<double> table;
    static const bool no_codes = getenv("BDF_NO_CODES") != nullptr;
    if (n_modes == 2 && nnz > 0 && !no_codes) {
        bool ok = true;
        for (int64_t i = 0; i < nnz && ok; i++) {
            const double v = values[i];
            auto it = std::lower_bound(table.begin(), table.end(), v);
            if (it != table.end() && *it == v) continue;
            if (!(v == v) || table.size() == 256) { ok = false; break; }       // NaN or too many
            table.insert(it, v);
        }
        if (!ok) table.clear();
    }
    std::vector<uint8_t> code_by_row;                                         // code of every observation, in table (COO) order
    if (!table.empty()) {
        code_by_row.resize((size_t)nnz);
        for (int64_t i = 0; i < nnz; i++)
            code_by_row[(size_t)i] = (uint8_t)(std::lower_bound(table.begin(), table.end(), values[i]) - table.begin());
        std::vector<double> t256(256, 0.0);
        std::copy(table.begin(), table.end(), t256.begin());
        int rct = upload(ctx, t256, &r->table_dev);
        if (rct) return rct;
        r->n_codes = (int)table.size();
    }

    {
        int64_t *rps[BDF_MAX_MODES], *ris[BDF_MAX_MODES];
        for (int m = 0; m < n_modes; m++) {
            r->idx[m].rowptr.assign((size_t)dims[m] + 1, 0);
            r->idx[m].rowids.assign((size_t)std::max<int64_t>(nnz, 1), 0);
            rps[m] = r->idx[m].rowptr.data();
            ris[m] = r->idx[m].rowids.data();
        }
        int rc = bdf_index_build(n_modes, dims, nnz, ids, id_bytes, rps, ris);
        if (rc) return rc;
    }
    for (int m = 0; m < n_modes; m++) {
        bdf_mode_index &ix = r->idx[m];
        const int64_t N = dims[m];
        ix.order.resize((size_t)N);
        std::iota(ix.order.begin(), ix.order.end(), 0);
        std::stable_sort(ix.order.begin(), ix.order.end(), [&](int32_t x, int32_t y) {
            return (ix.rowptr[(size_t)x + 1] - ix.rowptr[(size_t)x]) > (ix.rowptr[(size_t)y + 1] - ix.rowptr[(size_t)y]);
        });
        int rc;
        if (!sharded) {
            std::vector<int32_t> colidx((size_t)nnz * (size_t)(n_modes - 1) + 1, 0);       // (+1: the row kernel reads ids and values in pairs)
            std::vector<double> vals((size_t)nnz + 1, 0.0);
            std::vector<int32_t> perm((size_t)nnz);
            for (int64_t q = 0; q < nnz; q++) {
                int64_t i = ix.rowids[(size_t)q] - 1;
                perm[(size_t)q] = (int32_t)i;
                vals[(size_t)q] = values[i];
                int plane = 0;
                for (int k = 0; k < n_modes; k++) {
                    if (k == m) continue;
                    colidx[(size_t)plane * (size_t)nnz + (size_t)q] = (int32_t)(idat(i, k) - 1);
                    plane++;
                }
            }
            if ((rc = upload(ctx, ix.rowptr, &ix.rowptr_dev))) return rc;
            if ((rc = upload(ctx, colidx, &ix.colidx_dev))) return rc;
            if ((rc = upload(ctx, vals, &ix.vals_dev))) return rc;
            if ((rc = upload(ctx, perm, &ix.perm_dev))) return rc;
            if ((rc = upload(ctx, ix.order, &ix.order_dev))) return rc;
            if (r->n_codes && r->nint[1 - m] < (1 << 24)) {
                std::vector<uint32_t> packed((size_t)nnz + 1, 0u);          // (+1: the row kernel reads the words in pairs)
                for (int64_t q = 0; q < nnz; q++) packed[(size_t)q] = ((uint32_t)code_by_row[(size_t)perm[(size_t)q]] << 24) | (uint32_t)colidx[(size_t)q];
                if ((rc = upload(ctx, packed, &ix.packed_dev))) return rc;
            }
            ix.own_nnz = nnz;
            continue;
        }
        // the rows this rank owns, by internal position (chunk-major): position (c * world + rank) * cmax + i
        const int64_t cm = cmax[m];
        std::vector<int32_t> at((size_t)(cm * chunks), -1);         // owned slot -> original row
        for (int64_t row = 0; row < N; row++) {
            const int64_t p = pos[m][row];
            BDF_REQUIRE(p >= 0 && p < r->nint[m], BDF_ERR_ARG, "bdf_relation_create_sharded: position %lld of mode %d out of range", (long long)p, m);
            const int64_t blk = p / cm, c = blk / world;
            if (blk % world == rank) at[(size_t)(c * cm + p % cm)] = (int32_t)row;
        }
        ix.chunk_begin.assign((size_t)chunks + 1, 0);
        ix.own_q.assign(1, 0);
        for (int c = 0; c < chunks; c++) {
            ix.chunk_begin[(size_t)c] = (int64_t)ix.own_orig.size();
            for (int64_t i = 0; i < cm; i++) {
                const int32_t row = at[(size_t)(c * cm + i)];
                if (row < 0) continue;
                ix.own_orig.push_back(row);
                ix.own_pos.push_back((int32_t)(((int64_t)c * world + rank) * cm + i));
                ix.own_q.push_back(ix.own_q.back() + (ix.rowptr[(size_t)row + 1] - ix.rowptr[(size_t)row]));
            }
        }
        ix.chunk_begin[(size_t)chunks] = (int64_t)ix.own_orig.size();
        const int64_t on = ix.own_q.back();
        ix.own_nnz = on;
        std::vector<int32_t> colidx((size_t)on * (size_t)(n_modes - 1) + 1, 0);
        std::vector<double> vals((size_t)on + 1, 0.0);
        std::vector<int32_t> perm((size_t)on);
        for (size_t o = 0; o < ix.own_orig.size(); o++) {
            const int64_t row = ix.own_orig[o];
            int64_t dst = ix.own_q[o];
            for (int64_t q = ix.rowptr[(size_t)row]; q < ix.rowptr[(size_t)row + 1]; q++, dst++) {
                const int64_t i = ix.rowids[(size_t)q] - 1;
                perm[(size_t)dst] = (int32_t)i;
                vals[(size_t)dst] = values[i];
                int plane = 0;
                for (int k = 0; k < n_modes; k++) {
                    if (k == m) continue;
                    colidx[(size_t)plane * (size_t)on + (size_t)dst] = pos[k][idat(i, k) - 1];
                    plane++;
                }
            }
        }
        if ((rc = upload(ctx, colidx, &ix.colidx_dev))) return rc;
        if ((rc = upload(ctx, vals, &ix.vals_dev))) return rc;
        if ((rc = upload(ctx, perm, &ix.perm_dev))) return rc;
        if (r->n_codes && r->nint[1 - m] < (1 << 24)) {
            std::vector<uint32_t> packed((size_t)on + 1, 0u);
            for (int64_t q = 0; q < on; q++) packed[(size_t)q] = ((uint32_t)code_by_row[(size_t)perm[(size_t)q]] << 24) | (uint32_t)colidx[(size_t)q];
            if ((rc = upload(ctx, packed, &ix.packed_dev))) return rc;
        }
    }
    guard.r = nullptr;
    *out = r;
    return BDF_OK;
}

extern "C" int bdf_relation_create(bdf_ctx *ctx, int n_modes, const int64_t *dims, int64_t nnz,
                                   const void *ids, int id_bytes, const double *values, bdf_rel **out)
{
    return relation_create(ctx, n_modes, dims, nnz, ids, id_bytes, values, nullptr, nullptr, 0, 1, 1, out);
}

extern "C" int bdf_relation_create_sharded(bdf_ctx *ctx, int n_modes, const int64_t *dims, int64_t nnz, const void *ids,
                                           int id_bytes, const double *values, const int32_t *const *pos, const int64_t *cmax,
                                           int rank, int world, int chunks, bdf_rel **out)
{
    BDF_REQUIRE(pos, BDF_ERR_ARG, "bdf_relation_create_sharded: pos is NULL");
    return relation_create(ctx, n_modes, dims, nnz, ids, id_bytes, values, pos, cmax, rank, world, chunks, out);
}

extern "C" int bdf_relation_destroy(bdf_rel *rel)
{
    if (!rel) return BDF_OK;
    hipSetDevice(rel->ctx->device);
    hipStreamSynchronize(rel->ctx->stream);
    bdf_plans_release(rel->ctx, rel->serial);
    for (int m = 0; m < rel->n_modes; m++) {
        bdf_mode_index &ix = rel->idx[m];
        hipFree(ix.rowptr_dev); hipFree(ix.colidx_dev); hipFree(ix.vals_dev); hipFree(ix.perm_dev); hipFree(ix.order_dev);
        hipFree(ix.packed_dev);
    }
    hipFree(rel->table_dev);
    delete rel;
    return BDF_OK;
}

extern "C" int bdf_relation_index(const bdf_rel *rel, int mode, const int64_t **rowptr, const int64_t **rowids)
{
    BDF_REQUIRE(rel && rowptr && rowids, BDF_ERR_ARG, "bdf_relation_index: NULL argument");
    BDF_REQUIRE(mode >= 0 && mode < rel->n_modes, BDF_ERR_ARG, "bdf_relation_index: mode %d out of range", mode);
    *rowptr = rel->idx[mode].rowptr.data();
    *rowids = rel->idx[mode].rowids.data();
    return BDF_OK;
}

extern "C" int bdf_relation_value_mean(const bdf_rel *rel, double *mean)
{
    BDF_REQUIRE(rel && mean, BDF_ERR_ARG, "bdf_relation_value_mean: NULL argument");
    *mean = rel->value_mean;
    return BDF_OK;
}

extern "C" int bdf_relation_order(const bdf_rel *rel, int mode, int32_t *order_host)
{
    BDF_REQUIRE(rel && order_host, BDF_ERR_ARG, "bdf_relation_order: NULL argument");
    BDF_REQUIRE(mode >= 0 && mode < rel->n_modes, BDF_ERR_ARG, "bdf_relation_order: mode %d out of range", mode);
    memcpy(order_host, rel->idx[mode].order.data(), rel->idx[mode].order.size() * sizeof(int32_t));
    return BDF_OK;
}

// ---- a3-a7: rows ---------------------------------------------------------------------------
static int fill_args(bdf_ctx *ctx, const char *who, int D, int64_t N, int n_terms, const bdf_term *terms,
                     const double *mu, int mu_is_matrix, const double *Lambda, SampleArgs &a)
{
    BDF_REQUIRE(ctx && terms && mu && Lambda, BDF_ERR_ARG, "%s: NULL argument", who);
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "%s: num_latent=%d must be in 1..%d", who, D, BDF_MAX_D);
    BDF_REQUIRE(n_terms >= 1 && n_terms <= BDF_MAX_TERMS, BDF_ERR_ARG, "%s: n_terms=%d must be in 1..%d", who, n_terms, BDF_MAX_TERMS);
    memset(&a, 0, sizeof(a));
    for (int r = 0; r < n_terms; r++) {
        const bdf_term &t = terms[r];
        BDF_REQUIRE(t.rel != nullptr, BDF_ERR_ARG, "%s: terms[%d].rel is NULL", who, r);
        // a relation may be used from any context (stream) of the device it was created on
        BDF_REQUIRE(t.rel->ctx->device == ctx->device, BDF_ERR_ARG, "%s: terms[%d].rel lives on another device", who, r);
        BDF_REQUIRE(t.mode >= 0 && t.mode < t.rel->n_modes, BDF_ERR_ARG, "%s: terms[%d].mode=%d out of range", who, r, t.mode);
        BDF_REQUIRE(t.rel->nint[t.mode] == N, BDF_ERR_ARG,
                    "%s: entity has %lld instances, relation %d has data for %lld (ArgumentError)", who, (long long)N, r,
                    (long long)t.rel->nint[t.mode]);
        BDF_REQUIRE(r == 0 || t.rel->sharded == terms[0].rel->sharded, BDF_ERR_ARG, "%s: relations with and without a layout mixed", who);
        const bdf_mode_index &ix = t.rel->idx[t.mode];
        TermDev &T = a.t[r];
        T.rowptr = ix.rowptr_dev;
        T.colidx = ix.colidx_dev;
        T.vals = ix.vals_dev;
        T.perm = ix.perm_dev;
        T.linear = t.linear_values;
        T.nnz = ix.own_nnz;                 // plane stride of colidx: the observations held on this device
        T.n_other = t.rel->n_modes - 1;
        int plane = 0;
        bool lean = t.linear_values == nullptr && T.n_other <= 2, wide = false;
        for (int k = 0; k < t.rel->n_modes; k++) {
            if (k == t.mode) continue;
            BDF_REQUIRE(t.factors[k] != nullptr, BDF_ERR_ARG, "%s: terms[%d].factors[%d] is NULL", who, r, k);
            T.fac[plane++] = t.factors[k];
            lean = lean && t.rel->nint[k] < ((int64_t)1 << 32);
            wide = wide || !(t.rel->nint[k] < (1 << 24) && t.rel->nint[k] * (int64_t)D * 8 < ((int64_t)1 << 32));
        }
        T.lean = !lean ? 0 : (!wide ? 1 : (D > 32 ? 2 : 0));      // 2: 64-bit row offsets (compiled for D > 32 only)
        // parity hook (bdf_ctx_set_gather): force the general path / the 64-bit lean path (D > 32)
        if (ctx->gather_mode == 1) T.lean = 0;
        if (ctx->gather_mode == 2 && lean && D > 32) T.lean = 2;
        T.alpha = t.alpha;
        T.alpha_dev = t.alpha_dev;
        T.mean = t.mean_value;
        if (T.lean == 1 && T.n_other == 1 && ix.packed_dev) { T.packed = ix.packed_dev; T.table = t.rel->table_dev; T.n_codes = t.rel->n_codes; }
    }
    a.n_terms = n_terms;
    a.D = D;
    a.mu = mu;
    a.mu_is_matrix = mu_is_matrix;
    a.Lambda = Lambda;
    a.sweep = ctx->sweep_host;
    a.seed = ctx->seed;
    a.flag = ctx->flag_dev;
    return BDF_OK;
}

extern "C" int bdf_sample_rows(bdf_ctx *ctx, int D, int64_t N, int n_terms, const bdf_term *terms,
                               const double *mu, int mu_is_matrix, const double *Lambda,
                               uint32_t entity_tag, int shard, int n_shards, double *out, const double *prior_pack)
{
    SampleArgs a;
    int rc = fill_args(ctx, "bdf_sample_rows", D, N, n_terms, terms, mu, mu_is_matrix, Lambda, a);
    if (rc) return rc;
    BDF_REQUIRE(out != nullptr, BDF_ERR_ARG, "bdf_sample_rows: out is NULL");
    BDF_REQUIRE(n_shards >= 1 && shard >= 0 && shard < n_shards, BDF_ERR_ARG, "bdf_sample_rows: shard %d of %d", shard, n_shards);
    if (terms[0].rel->sharded) {
        // a relation created with a layout holds this rank's rows only: (shard, n_shards) = (chunk, chunks)
        BDF_REQUIRE(n_shards == terms[0].rel->chunks, BDF_ERR_ARG, "bdf_sample_rows: the relation was created with %d chunks: pass (chunk, %d)",
                    terms[0].rel->chunks, terms[0].rel->chunks);
        for (int r = 1; r < n_terms; r++)
            BDF_REQUIRE(terms[r].rel->sharded && terms[r].rel->chunks == terms[0].rel->chunks && terms[r].rel->rank == terms[0].rel->rank &&
                            terms[r].rel->idx[terms[r].mode].own_orig == terms[0].rel->idx[terms[0].mode].own_orig,
                        BDF_ERR_ARG, "bdf_sample_rows: the entity's relations were created with different layouts");
    }
    const bdf_rel *rels[BDF_MAX_TERMS];
    int modes[BDF_MAX_TERMS];
    for (int r = 0; r < n_terms; r++) {
        rels[r] = terms[r].rel;
        modes[r] = terms[r].mode;
        for (int k = 0; k < terms[r].rel->n_modes; k++)
            BDF_REQUIRE(k == terms[r].mode || terms[r].factors[k] != out, BDF_ERR_ARG,
                        "bdf_sample_rows: out aliases terms[%d].factors[%d]", r, k);
    }
    a.entity_tag = entity_tag;
    a.out = out;
    if (prior_pack && !mu_is_matrix) { a.prior_b = prior_pack; a.prior_c = prior_pack + D; }
    if (ctx->rows_ready && prior_pack && !mu_is_matrix) { a.ready = ctx->rows_ready; a.ready_want = ctx->rows_ready_want; }
    ctx->rows_ready = nullptr;
    a.span = ctx->rows_span;
    ctx->rows_span = nullptr;
    a.done = ctx->rows_done;
    ctx->rows_done = nullptr;
    ctx->rows_done_added = -1;
#ifdef BDF_K1_SPANS
    {   // diagnostic build only: a ring of 1024 launches x 8192 waves x {start, end, wait} (bdf_debug_spans)
        extern unsigned long long *g_bdf_span_buf;
        extern unsigned long long g_bdf_span_count;
        if (!g_bdf_span_buf) {
            BDF_HIP(hipMalloc((void **)&g_bdf_span_buf, (size_t)1024 * 8192 * 3 * 8));
            BDF_HIP(hipMemset(g_bdf_span_buf, 0, (size_t)1024 * 8192 * 3 * 8));
            BDF_HIP(hipDeviceSynchronize());
        }
        a.b_dump = (double *)(g_bdf_span_buf + (size_t)8192 * 3 * (g_bdf_span_count++ % 1024));
    }
#endif
#ifdef BDF_K1_STAMPS
    {   // diagnostic build only: per-wave phase stamps (16 x u64 per wave) readable through bdf_debug_stamps
        extern void *g_bdf_stamp_buf;
        if (!g_bdf_stamp_buf) { BDF_HIP(hipMalloc(&g_bdf_stamp_buf, 16 * 8 * 65536)); }
        BDF_HIP(hipMemsetAsync(g_bdf_stamp_buf, 0, 16 * 8 * 65536, ctx->stream));
        a.b_dump = (double *)g_bdf_stamp_buf;
    }
#endif
    return bdf_launch_sample_rows(ctx, a, rels, modes, shard, n_shards, false);
}

#ifdef BDF_K1_SPANS
unsigned long long *g_bdf_span_buf = nullptr;
unsigned long long g_bdf_span_count = 0;
extern "C" int bdf_debug_spans(unsigned long long *host, unsigned long long *count)
{
    BDF_HIP(hipDeviceSynchronize());
    if (g_bdf_span_buf) BDF_HIP(hipMemcpy(host, g_bdf_span_buf, (size_t)1024 * 8192 * 3 * 8, hipMemcpyDeviceToHost));
    *count = g_bdf_span_count;
    return BDF_OK;
}
#endif
#ifdef BDF_K1_STAMPS
void *g_bdf_stamp_buf = nullptr;
extern "C" int bdf_debug_stamps(bdf_ctx *ctx, unsigned long long *host, int nwaves)
{
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    BDF_HIP(hipMemcpy(host, g_bdf_stamp_buf, (size_t)nwaves * 16 * 8, hipMemcpyDeviceToHost));
    return BDF_OK;
}
#endif

extern "C" int bdf_row_system(bdf_ctx *ctx, int D, int64_t N, int n_terms, const bdf_term *terms,
                              const double *mu, int mu_is_matrix, const double *Lambda,
                              double *P_out, double *b_out)
{
    SampleArgs a;
    int rc = fill_args(ctx, "bdf_row_system", D, N, n_terms, terms, mu, mu_is_matrix, Lambda, a);
    if (rc) return rc;
    BDF_REQUIRE(P_out && b_out, BDF_ERR_ARG, "bdf_row_system: NULL output");
    const bdf_rel *rels[BDF_MAX_TERMS];
    int modes[BDF_MAX_TERMS];
    for (int r = 0; r < n_terms; r++) { rels[r] = terms[r].rel; modes[r] = terms[r].mode; }
    a.P_dump = P_out;
    a.b_dump = b_out;
    return bdf_launch_sample_rows(ctx, a, rels, modes, 0, 1, true);
}

__global__ void k_normals(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
                          int64_t row_begin, int64_t n_rows, int n, double *out)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * n) return;
    int64_t r = idx / n;
    int e = (int)(idx % n);
    out[idx] = bdf_normal(seed, sweep, purpose, entity, (uint64_t)(row_begin + r), e);
}

extern "C" int bdf_normals(bdf_ctx *ctx, uint32_t purpose, uint32_t entity_tag, int64_t row_begin,
                           int64_t n_rows, int n, double *out)
{
    BDF_REQUIRE(ctx && out && n >= 1 && n_rows >= 0, BDF_ERR_ARG, "bdf_normals: bad argument");
    int64_t total = n_rows * n;
    if (total == 0) return BDF_OK;
    hipLaunchKernelGGL(k_normals, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->seed,
                       ctx->sweep_host, purpose, entity_tag, row_begin, n_rows, n, out);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

__global__ void k_philox(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity, uint64_t row,
                         uint32_t pair, uint32_t *out)
{
    u32x4 o = bdf_draw(seed, sweep, purpose, entity, row, pair);
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w;
}

extern "C" int bdf_philox(bdf_ctx *ctx, uint32_t purpose, uint32_t entity_tag, uint64_t row, uint32_t pair,
                          uint32_t out_host[4])
{
    BDF_REQUIRE(ctx && out_host, BDF_ERR_ARG, "bdf_philox: NULL argument");
    void *s;
    int rc = bdf_scratch(ctx, 256, &s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_philox, dim3(1), dim3(1), 0, ctx->stream, ctx->seed, ctx->sweep_host, purpose, entity_tag, row,
                       pair, (uint32_t *)s);
    BDF_HIP(hipGetLastError());
    return bdf_d2h(ctx, out_host, s, 4 * sizeof(uint32_t));
}
