// bdf_api.hip -- C-ABI entry points: context, device memory, IndexedDF -> device CSR, row sampling front-end
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
    c->device = device;
    c->seed = seed;
    c->own_stream = false;
    c->stream = (hipStream_t)stream;          // NULL is the device's default stream
    BDF_HIP(hipMalloc((void **)&c->sweep_dev, sizeof(uint32_t)));
    BDF_HIP(hipMalloc((void **)&c->flag_dev, 16 * sizeof(int)));      // [0] error flag, [1..] self-resetting arrival counters
    BDF_HIP(hipMemsetAsync(c->sweep_dev, 0, sizeof(uint32_t), c->stream));
    BDF_HIP(hipMemsetAsync(c->flag_dev, 0, 16 * sizeof(int), c->stream));
    BDF_HIP(hipMalloc((void **)&c->rows_done_dev, BDF_GATE_COUNTERS * BDF_GATE_STRIDE * sizeof(uint32_t)));
    BDF_HIP(hipMemsetAsync(c->rows_done_dev, 0, BDF_GATE_COUNTERS * BDF_GATE_STRIDE * sizeof(uint32_t), c->stream));
    memset(c->rows_done_target, 0, sizeof(c->rows_done_target));
    c->time_start = c->time_stop = nullptr;
    c->time_h_start = c->time_h_stop = nullptr;
    c->time_gate_stop = nullptr;
    c->sweep_host = 0;
    c->skip_flag = nullptr;
    c->cg_status = nullptr;
    c->cg_gen = 0;
    c->scratch = nullptr;
    c->scratch2 = nullptr;
    c->scratch2_bytes = 0;
    c->scratch_bytes = 0;
    c->item_size = 192;
    c->piece_size = 128;
    {
        const char *force = getenv("BDF_GATHER");
        c->gather_mode = force && !strcmp(force, "general") ? 1 : (force && !strcmp(force, "wide") ? 2 : 0);
    }
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
    hipFree(ctx->sweep_dev);
    hipFree(ctx->flag_dev);
    hipFree(ctx->rows_done_dev);
    if (ctx->scratch2) hipFree(ctx->scratch2);
    if (ctx->cg_status) hipHostFree((void *)ctx->cg_status);
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
    BDF_REQUIRE(ctx && observations >= 8 && observations <= (1 << 20), BDF_ERR_ARG, "bdf_ctx_set_item_size: 8..2^20 observations");
    ctx->item_size = observations;
    ctx->piece_size = std::max(8, observations * 2 / 3);
    return BDF_OK;
}

extern "C" int bdf_ctx_set_gather(bdf_ctx *ctx, int mode)
{
    BDF_REQUIRE(ctx && mode >= 0 && mode <= 2, BDF_ERR_ARG, "bdf_ctx_set_gather: mode must be 0 (auto), 1 (general) or 2 (64-bit offsets)");
    ctx->gather_mode = mode;
    return BDF_OK;
}

extern "C" int bdf_ctx_set_piece_size(bdf_ctx *ctx, int observations)
{
    BDF_REQUIRE(ctx && observations >= 8 && observations <= ctx->item_size, BDF_ERR_ARG,
                "bdf_ctx_set_piece_size: 8..item size (%d) observations", ctx ? ctx->item_size : 0);
    ctx->piece_size = observations;
    return BDF_OK;
}

extern "C" int bdf_ctx_sync(bdf_ctx *ctx)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_ctx_sync: ctx is NULL");
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    int flag = 0;
    BDF_HIP(hipMemcpy(&flag, ctx->flag_dev, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
        BDF_HIP(hipMemset(ctx->flag_dev, 0, sizeof(int)));
        if (flag & 32) {
            bdf_set_error("bdf_rows_gate: timed out waiting for the row kernels of the other context (flag %d)", flag);
            return BDF_ERR_HIP;
        }
        bdf_set_error("a matrix that must be positive definite was not (flag %d)", flag);
        return BDF_ERR_NOTPD;
    }
    return BDF_OK;
}

// ---- cross-stream hand-over without events: a one-wave gate kernel on the waiting stream ---------------------------
// hipEventRecord after a row kernel + hipStreamWaitEvent costs the recording stream ~9 us per launch (tools/event_cost3.hip);
// completion counters written by the row kernel itself and a gate kernel polling them on the waiting stream cost it ~2.5.
namespace {
struct GateTargets { uint32_t t[BDF_GATE_COUNTERS]; };

__global__ __launch_bounds__(64) void k_rows_gate(const uint32_t *counters, GateTargets tg, int *flag, long long max_ticks)
{
    const int lane = threadIdx.x;
    const long long t0 = wall_clock64();                       // 100 MHz
    const uint32_t want = tg.t[lane];
    for (;;) {
        const uint32_t c = __hip_atomic_load(counters + lane * BDF_GATE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64((int32_t)(c - want) < 0) == 0) break;
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > max_ticks) {                 // bounded: a bug must not hang the device
            if (lane == 0) atomicOr(flag, 32);
            break;
        }
    }
}

__global__ void k_gate_bump(uint32_t *counters)
{
    if (threadIdx.x < BDF_GATE_COUNTERS)
        __hip_atomic_fetch_add(counters + threadIdx.x * BDF_GATE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

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

extern "C" int bdf_rows_gate(bdf_ctx *waiter, const bdf_ctx *producer)
{
    BDF_REQUIRE(waiter && producer, BDF_ERR_ARG, "bdf_rows_gate: NULL context");
    BDF_REQUIRE(waiter->device == producer->device, BDF_ERR_ARG, "bdf_rows_gate: the contexts are on different devices");
    if (waiter->stream == producer->stream) return BDF_OK;     // same stream: already ordered
    GateTargets tg;
    memcpy(tg.t, producer->rows_done_target, sizeof(tg.t));
    static const long long max_ticks = 100000000LL * (getenv("BDF_GATE_TIMEOUT_S") ? atoll(getenv("BDF_GATE_TIMEOUT_S")) : 30);
    hipLaunchKernelGGL(k_rows_gate, dim3(1), dim3(64), 0, waiter->stream, producer->rows_done_dev, tg, waiter->flag_dev, max_ticks);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

extern "C" int bdf_gate_snapshot(const bdf_ctx *producer, uint32_t *targets)
{
    BDF_REQUIRE(producer && targets, BDF_ERR_ARG, "bdf_gate_snapshot: NULL argument");
    memcpy(targets, producer->rows_done_target, sizeof(producer->rows_done_target));
    return BDF_OK;
}

extern "C" int bdf_rows_gate_at(bdf_ctx *waiter, const bdf_ctx *producer, const uint32_t *targets)
{
    BDF_REQUIRE(waiter && producer && targets, BDF_ERR_ARG, "bdf_rows_gate_at: NULL argument");
    BDF_REQUIRE(waiter->device == producer->device, BDF_ERR_ARG, "bdf_rows_gate_at: the contexts are on different devices");
    if (waiter->stream == producer->stream) return BDF_OK;
    GateTargets tg;
    memcpy(tg.t, targets, sizeof(tg.t));
    static const long long max_ticks = 100000000LL * (getenv("BDF_GATE_TIMEOUT_S") ? atoll(getenv("BDF_GATE_TIMEOUT_S")) : 30);
    // (the end of the gate, not the start event of the kernel behind it, is when that kernel can begin: an event attached
    // to a dispatch that waits behind a spinning gate is stamped while it waits)
    hipExtLaunchKernelGGL(k_rows_gate, dim3(1), dim3(64), 0, waiter->stream, nullptr, waiter->time_gate_stop, 0,
                          producer->rows_done_dev, tg, waiter->flag_dev, max_ticks);
    waiter->time_gate_stop = nullptr;
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

extern "C" int bdf_ctx_time_next_gate(bdf_ctx *ctx, void *stop)
{
    BDF_REQUIRE(ctx, BDF_ERR_ARG, "bdf_ctx_time_next_gate: ctx is NULL");
    ctx->time_gate_stop = (hipEvent_t)stop;
    return BDF_OK;
}

extern "C" int bdf_rows_gate_selftest(bdf_ctx *waiter, bdf_ctx *producer, int *usable)
{
    BDF_REQUIRE(waiter && producer && usable, BDF_ERR_ARG, "bdf_rows_gate_selftest: NULL argument");
    *usable = 0;
    if (waiter->device != producer->device) return BDF_OK;
    if (waiter->stream == producer->stream) { *usable = 1; return BDF_OK; }
    // the gate is enqueued FIRST and the kernel that satisfies it afterwards on the other stream: if the two streams
    // cannot run side by side (they share a hardware queue), the gate times out (20 ms) instead of passing
    BDF_HIP(hipStreamSynchronize(producer->stream));
    BDF_HIP(hipStreamSynchronize(waiter->stream));
    GateTargets tg;
    for (int c = 0; c < BDF_GATE_COUNTERS; c++) tg.t[c] = ++producer->rows_done_target[c];
    hipLaunchKernelGGL(k_rows_gate, dim3(1), dim3(64), 0, waiter->stream, producer->rows_done_dev, tg, waiter->flag_dev, 2000000LL);
    hipLaunchKernelGGL(k_gate_bump, dim3(1), dim3(64), 0, producer->stream, producer->rows_done_dev);
    BDF_HIP(hipGetLastError());
    BDF_HIP(hipStreamSynchronize(waiter->stream));
    BDF_HIP(hipStreamSynchronize(producer->stream));
    int flag = 0;
    BDF_HIP(hipMemcpy(&flag, waiter->flag_dev, sizeof(int), hipMemcpyDeviceToHost));
    if (flag & 32) {
        flag &= ~32;
        BDF_HIP(hipMemcpy(waiter->flag_dev, &flag, sizeof(int), hipMemcpyHostToDevice));
    } else {
        *usable = 1;
    }
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

extern "C" int bdf_relation_create(bdf_ctx *ctx, int n_modes, const int64_t *dims, int64_t nnz,
                                   const void *ids, int id_bytes, const double *values, bdf_rel **out)
{
    BDF_REQUIRE(ctx && dims && out, BDF_ERR_ARG, "bdf_relation_create: NULL argument");
    BDF_REQUIRE(n_modes >= 2 && n_modes <= BDF_MAX_MODES, BDF_ERR_ARG,
                "bdf_relation_create: n_modes=%d must be in 2..%d", n_modes, BDF_MAX_MODES);
    BDF_REQUIRE(id_bytes == 4 || id_bytes == 8, BDF_ERR_ARG, "bdf_relation_create: id_bytes must be 4 or 8");
    BDF_REQUIRE(nnz >= 0 && nnz < (int64_t)0x7fffffff, BDF_ERR_ARG, "bdf_relation_create: nnz=%lld unsupported", (long long)nnz);
    BDF_REQUIRE(nnz == 0 || (ids && values), BDF_ERR_ARG, "bdf_relation_create: ids/values NULL");
    for (int m = 0; m < n_modes; m++)
        BDF_REQUIRE(dims[m] >= 0 && dims[m] < (int64_t)0x7fffffff, BDF_ERR_ARG, "bdf_relation_create: dims[%d]=%lld unsupported", m, (long long)dims[m]);
    BDF_HIP(hipSetDevice(ctx->device));
    auto idat = [&](int64_t i, int m) -> int64_t { return id_at(ids, id_bytes, nnz, i, m); };

    static std::atomic<uint64_t> next_serial{1};
    bdf_rel *r = new bdf_rel();
    r->ctx = ctx;
    r->serial = next_serial.fetch_add(1);
    r->n_modes = n_modes;
    r->nnz = nnz;
    for (int m = 0; m < n_modes; m++) r->dims[m] = dims[m];
    double s = 0.0;
    for (int64_t i = 0; i < nnz; i++) s += values[i];
    r->value_mean = nnz ? s / (double)nnz : NAN;

    {
        int64_t *rps[BDF_MAX_MODES], *ris[BDF_MAX_MODES];
        for (int m = 0; m < n_modes; m++) {
            r->idx[m].rowptr.assign((size_t)dims[m] + 1, 0);
            r->idx[m].rowids.assign((size_t)std::max<int64_t>(nnz, 1), 0);
            rps[m] = r->idx[m].rowptr.data();
            ris[m] = r->idx[m].rowids.data();
        }
        int rc = bdf_index_build(n_modes, dims, nnz, ids, id_bytes, rps, ris);
        if (rc) { delete r; return rc; }
    }
    for (int m = 0; m < n_modes; m++) {
        bdf_mode_index &ix = r->idx[m];
        const int64_t N = dims[m];
        std::vector<int32_t> colidx((size_t)nnz * (size_t)(n_modes - 1));
        std::vector<double> vals((size_t)nnz);
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
        ix.order.resize((size_t)N);
        std::iota(ix.order.begin(), ix.order.end(), 0);
        std::stable_sort(ix.order.begin(), ix.order.end(), [&](int32_t x, int32_t y) {
            return (ix.rowptr[(size_t)x + 1] - ix.rowptr[(size_t)x]) > (ix.rowptr[(size_t)y + 1] - ix.rowptr[(size_t)y]);
        });
        int rc;
        if ((rc = upload(ctx, ix.rowptr, &ix.rowptr_dev))) return rc;
        if ((rc = upload(ctx, colidx, &ix.colidx_dev))) return rc;
        if ((rc = upload(ctx, vals, &ix.vals_dev))) return rc;
        if ((rc = upload(ctx, perm, &ix.perm_dev))) return rc;
        if ((rc = upload(ctx, ix.order, &ix.order_dev))) return rc;
    }
    *out = r;
    return BDF_OK;
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
    }
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
        BDF_REQUIRE(t.rel->dims[t.mode] == N, BDF_ERR_ARG,
                    "%s: entity has %lld instances, relation %d has data for %lld (ArgumentError)", who, (long long)N, r,
                    (long long)t.rel->dims[t.mode]);
        const bdf_mode_index &ix = t.rel->idx[t.mode];
        TermDev &T = a.t[r];
        T.rowptr = ix.rowptr_dev;
        T.colidx = ix.colidx_dev;
        T.vals = ix.vals_dev;
        T.perm = ix.perm_dev;
        T.linear = t.linear_values;
        T.nnz = t.rel->nnz;
        T.n_other = t.rel->n_modes - 1;
        int plane = 0;
        bool lean = t.linear_values == nullptr && T.n_other <= 2, wide = false;
        for (int k = 0; k < t.rel->n_modes; k++) {
            if (k == t.mode) continue;
            BDF_REQUIRE(t.factors[k] != nullptr, BDF_ERR_ARG, "%s: terms[%d].factors[%d] is NULL", who, r, k);
            T.fac[plane++] = t.factors[k];
            lean = lean && t.rel->dims[k] < ((int64_t)1 << 32);
            wide = wide || !(t.rel->dims[k] < (1 << 24) && t.rel->dims[k] * (int64_t)D * 8 < ((int64_t)1 << 32));
        }
        T.lean = !lean ? 0 : (!wide ? 1 : (D > 32 ? 2 : 0));      // 2: 64-bit row offsets (compiled for D > 32 only)
        // parity hook (bdf_ctx_set_gather): force the general path / the 64-bit lean path (D > 32)
        if (ctx->gather_mode == 1) T.lean = 0;
        if (ctx->gather_mode == 2 && lean && D > 32) T.lean = 2;
        T.alpha = t.alpha;
        T.mean = t.mean_value;
    }
    a.n_terms = n_terms;
    a.D = D;
    a.mu = mu;
    a.mu_is_matrix = mu_is_matrix;
    a.Lambda = Lambda;
    a.sweep = ctx->sweep_host;
    a.seed = ctx->seed;
    a.flag = ctx->flag_dev;
    a.done = getenv("BDF_EXP_NO_DONE") ? nullptr : ctx->rows_done_dev;
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
