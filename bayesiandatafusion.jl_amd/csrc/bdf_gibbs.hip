// bdf_gibbs.hip -- one Gibbs iteration of macau (src/macau.jl:80-203) enqueued from native code: the latent rows of every
// entity (sample_latent_all2! / sample_user2_all!, src/sampling.jl:149-289), the exchange of the sampled rows between GPUs,
// every entity's hyperprior (src/sampling.jl:116-127, src/normal_wishart.jl:38-42) and the test-set prediction update
// (macau.jl:142-184) -- the BPMF iteration and the one the multi-GPU path runs.  (Side information adds the beta update and
// per-row prior means; those iterations are enqueued step by step through the entry points of bdf.h.)
//
// Three HIP streams: rows | hyperpriors | prediction updates.  The hyperprior of entity j runs beside the rows of entity
// j+1, the prediction update of sweep t beside the rows of sweep t+1 (every entity's rows rotate through three buffers).
// Hand-overs towards the hyperprior and prediction streams are HIP events that ride on the producing kernel's own dispatch
// packet (hipExtLaunchKernelGGL stop events): no marker packets on the row stream.  The hand-over BACK to the row stream --
// (mu, Lambda) of the draw, needed by the entity's next row launch -- is, by default, not an event: the draw runs on CUs
// the row kernels never occupy (bdf_ctx_create_rows), publishes a per-entity epoch when its prior pack is written, and
// the row kernel's waves poll that word where they first need the prior (BDF_NO_POLL=1, a profiler that serialises the
// streams, or several ranks: event waits instead).
#include "bdf_common.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

struct bdf_gibbs {
    bdf_ctx *rows, *hyper, *pred;        // rows: the caller's context; hyper / pred: owned (own streams)
    int D;
    struct Ent {
        bdf_gibbs_entity d;
        int cur;                         // buffer holding the current rows
        hipEvent_t ev_rows = nullptr, ev_hyper = nullptr;    // rows of this sweep complete (and exchanged) | (mu, Lambda) of this sweep complete
        hipEvent_t ev_beta = nullptr;                        // side information: beta (and lambda_beta) of this sweep complete
        bool beta_recorded = false;
        bool hyper_recorded = false;
        // number of hyperprior draws enqueued for this entity so far: the value its next draw publishes in ready_dev and the
        // entity's next row launch polls for.  Private and strictly increasing -- NOT the caller's sweep number, which may
        // repeat or restart (a repeated number would let the poll pass while the draw is still writing the pack)
        uint32_t epoch = 0;
        // the rows' hand-over to the chain by counter (no event): the sum the 64 words of done_dev reach when every row launch of
        // this entity enqueued so far has completed (wraps; compared as a difference)
        uint32_t *done_dev = nullptr;
        uint32_t done_target = 0;
        hipEvent_t t_start = nullptr, t_stop = nullptr;      // bdf_gibbs_time_rows: attached to the next row launch of this entity
        unsigned long long *span = nullptr;                  // bdf_gibbs_span_rows: SampleArgs::span of the next row launch of this entity
    };
    std::vector<Ent> ent;
    bdf_pairs *test;
    int32_t test_entity[BDF_MAX_MODES];
    double test_mean, clamp_lo, clamp_hi, class_cut;
    double *stats_dev;
    hipEvent_t ev_pred[3] = {nullptr, nullptr, nullptr};   // prediction update number k complete: ev_pred[k % 3]
    uint64_t pred_at[3] = {0, 0, 0};     // ... and the iteration count (n_iter) at which it was enqueued
    uint64_t n_pred;                     // prediction updates enqueued so far
    uint64_t n_iter = 0;                 // iterations enqueued so far (with or without a prediction update)
    bdf_comm *comm;                      // nullable: exchange of the sampled rows between the ranks after every entity
    std::vector<bdf_gibbs_relation> rels; // relations with a model of their own (alpha sampled, relation features): bdf_gibbs_set_relations
    double *rel_sse = nullptr;           // dev, 8 doubles: the squared-error statistics of sample_alpha
    // The row stream has NO wait for the hyperprior draws when they run on reserved CUs (bdf_ctx_create_rows): a draw there
    // cannot be starved by the chip-filling row kernel, so the row kernel is enqueued right behind its predecessor (back-to-back
    // row kernels start with no gap; a wait for another stream's event costs the row stream ~10 us even when long satisfied:
    // profiles/r02_sweep_timeline_events.txt) and its waves poll ready[entity] where they first need the prior.
    uint32_t *ready_dev;                 // per entity: iteration number of the last completed draw
    bool polling;
    // BDF_DEBUG: where the host's time in bdf_gibbs_sweep goes (waiting for the prediction update of two sweeps ago = the device
    // is the bottleneck; enqueueing = the host is)
    bool debug;
    double host_wait_us, host_enqueue_us;
    uint64_t n_sweeps;
};

namespace {

// do kernels on the two streams run side by side?  HIP multiplexes streams onto a few hardware queues; two streams on one
// queue serialise whatever the hand-over mechanism.  Timing test: a 60 us spin on each, together.
__global__ void k_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Can a kernel on `b` run WHILE a kernel on `a` is running?  (Not under profilers that serialise kernels across queues --
// rocprofv3 --pmc does -- nor if the two streams share a hardware queue.)  A kernel on `a` spins until a flag is set by a
// kernel enqueued on `b` afterwards; bounded at 20 ms.  The row kernels poll for the hyperprior draw only if this holds.
__global__ void k_wait_flag(const uint32_t *flag, long long max_ticks, uint32_t *timed_out)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > max_ticks) { *timed_out = 1; break; }
    }
}
__global__ void k_set_flag(uint32_t *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

int streams_concurrent(hipStream_t a, hipStream_t b, bool *yes)
{
    uint32_t *d;
    BDF_HIP(hipMalloc((void **)&d, 2 * sizeof(uint32_t)));
    struct Free { uint32_t *p; ~Free() { (void)hipFree(p); } } guard{d};
    BDF_HIP(hipMemsetAsync(d, 0, 2 * sizeof(uint32_t), a));
    BDF_HIP(hipStreamSynchronize(a)); BDF_HIP(hipStreamSynchronize(b));
    hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(1), 0, a, (const uint32_t *)d, 2000000LL, d + 1);
    hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, b, d);
    BDF_HIP(hipStreamSynchronize(a)); BDF_HIP(hipStreamSynchronize(b));
    uint32_t h[2] = {0, 0};
    BDF_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    *yes = h[1] == 0;
    return BDF_OK;
}

int streams_overlap(hipStream_t a, hipStream_t b, bool *yes)
{
    hipEvent_t e0, e1, e2;
    BDF_HIP(hipEventCreate(&e0)); BDF_HIP(hipEventCreate(&e1)); BDF_HIP(hipEventCreate(&e2));
    BDF_HIP(hipStreamSynchronize(a)); BDF_HIP(hipStreamSynchronize(b));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        BDF_HIP(hipEventRecord(e0, a));
        BDF_HIP(hipStreamWaitEvent(b, e0, 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 6000LL);       // 100 MHz clock: 60 us
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, 6000LL);
        BDF_HIP(hipEventRecord(e1, b));
        BDF_HIP(hipStreamWaitEvent(a, e1, 0));
        BDF_HIP(hipEventRecord(e2, a));
        BDF_HIP(hipEventSynchronize(e2));
        float ms = 0.f;
        BDF_HIP(hipEventElapsedTime(&ms, e0, e2));
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    *yes = best < 0.100f;               // side by side: ~0.065 ms; serialised: ~0.125 ms
    return BDF_OK;
}

// A non-blocking stream, optionally confined to a set of CUs.  CU mask bits are interleaved over the XCDs (bits 0..7 are CU 0
// of shader engine 0 of XCD 0..7: tools/cu_mask_probe.hip), so `reserve` = 8 k bits set aside k CUs in every XCD -- a kernel's
// workgroups are dealt over the XCDs, every XCD must have a CU of the mask.  reserved = true: the stream runs on those CUs
// only; false: on all the others.
// The streams live in a process-wide pool and are never destroyed: host frameworks keep references to a stream they have
// used (torch: allocator pools per stream, events recorded on it) beyond the life of the context that ran on it, and
// destroying it under them crashes at the framework's own teardown.  Slot k of (device, reserve, role) is always the same
// stream, so later engines of a process pick the same streams as the first.
int pooled_stream(int device, int reserve, bool reserved, int slot, hipStream_t *out)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, hipStream_t> pool;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(device, reserve, reserved ? 1 : 0, slot);
    auto it = pool.find(key);
    if (it != pool.end()) { *out = it->second; return BDF_OK; }
    hipStream_t st;
    if (reserve <= 0) {
        BDF_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    } else {
        hipDeviceProp_t prop;
        BDF_HIP(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mask((size_t)words, 0u);
        for (int i = 0; i < ncu; i++)
            if ((i < reserve) == reserved) mask[(size_t)(i / 32)] |= 1u << (i % 32);
        BDF_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()));
    }
    pool[key] = st;
    *out = st;
    return BDF_OK;
}

int make_side_ctx(bdf_ctx *main, const std::vector<bdf_ctx *> &apart, bool reserved, bdf_ctx **out)
{
    // a few candidate streams; the first that overlaps with the row stream and with every stream in `apart`
    bdf_ctx *fallback = nullptr;
    for (int attempt = 0; attempt < 8; attempt++) {
        hipStream_t st;
        // slot 0 of the unreserved role is the row stream itself
        int rc = pooled_stream(main->device, main->reserve_cus, reserved, attempt + 1, &st);
        if (rc) return rc;
        if (st == main->stream) continue;
        bool taken = false;
        for (bdf_ctx *o : apart) taken = taken || o->stream == st;
        if (taken) continue;
        bdf_ctx *c;
        rc = bdf_ctx_create(main->device, (void *)st, main->seed, &c);
        if (rc) return rc;
        c->reserve_cus = main->reserve_cus;
        c->on_reserved = (reserved && main->reserve_cus > 0) ? 1 : 0;
        bool ok = true;
        static const bool no_test = getenv("BDF_NO_STREAM_TEST") != nullptr;
        if (!no_test) {
            if ((rc = streams_overlap(st, main->stream, &ok))) return rc;
            for (bdf_ctx *o : apart)
                if (ok && (rc = streams_overlap(st, o->stream, &ok))) return rc;
        }
        if (ok) {
            if (fallback) bdf_ctx_destroy(fallback);
            *out = c;
            return BDF_OK;
        }
        if (!fallback) fallback = c; else bdf_ctx_destroy(c);
    }
    *out = fallback;                    // no candidate runs beside the others: correct, only slower
    return BDF_OK;
}

}  // namespace

extern "C" int bdf_ctx_create_rows(int device, uint64_t seed, int reserve_cus, bdf_ctx **out)
{
    BDF_REQUIRE(out && reserve_cus >= 0 && reserve_cus % 8 == 0 && reserve_cus <= 64, BDF_ERR_ARG,
                "bdf_ctx_create_rows: reserve_cus must be 0, 8, 16, ... 64 (whole CUs per XCD)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        bdf_set_error("bdf_ctx_create_rows: no HIP device visible (this library has no CPU path)");
        return BDF_ERR_NOGPU;
    }
    BDF_REQUIRE(device >= 0 && device < ndev, BDF_ERR_ARG, "bdf_ctx_create_rows: device %d out of range", device);
    BDF_HIP(hipSetDevice(device));
    hipStream_t st;
    int rc = pooled_stream(device, reserve_cus, false, 0, &st);
    if (rc) return rc;
    bdf_ctx *c;
    if ((rc = bdf_ctx_create(device, (void *)st, seed, &c))) return rc;
    c->reserve_cus = reserve_cus;
    *out = c;
    return BDF_OK;
}

extern "C" int bdf_ctx_create_side(bdf_ctx *main_ctx, bdf_ctx *const *apart, int n_apart, int reserved, bdf_ctx **out)
{
    BDF_REQUIRE(main_ctx && out && n_apart >= 0 && (n_apart == 0 || apart), BDF_ERR_ARG, "bdf_ctx_create_side: bad argument");
    std::vector<bdf_ctx *> ap(apart, apart + n_apart);
    return make_side_ctx(main_ctx, ap, reserved != 0, out);
}

extern "C" int bdf_gibbs_create(bdf_ctx *rows_ctx, int D, int n_entities, const bdf_gibbs_entity *ents, bdf_gibbs **out)
{
    BDF_REQUIRE(rows_ctx && ents && out, BDF_ERR_ARG, "bdf_gibbs_create: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_gibbs_create: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    BDF_REQUIRE(n_entities >= 1 && n_entities <= 64, BDF_ERR_ARG, "bdf_gibbs_create: 1..64 entities");
    for (int j = 0; j < n_entities; j++) {
        const bdf_gibbs_entity &e = ents[j];
        BDF_REQUIRE(e.n_terms >= 1 && e.n_terms <= BDF_MAX_TERMS, BDF_ERR_ARG, "bdf_gibbs_create: entity %d takes part in %d relations (1..%d)", j, e.n_terms, BDF_MAX_TERMS);
        BDF_REQUIRE(e.sample[0] && e.sample[1] && e.sample[2] && e.mu && e.Lambda && e.mu0 && e.WI && e.sumU && e.UUt && e.prior_pack && e.draws,
                    BDF_ERR_ARG, "bdf_gibbs_create: entity %d has a NULL buffer", j);
        BDF_REQUIRE(!e.feat || (e.beta && e.uhat && e.mu_matrix && e.Tinv && e.lambda_beta), BDF_ERR_ARG,
                    "bdf_gibbs_create: entity %d has side information but a NULL beta / uhat / mu_matrix / Tinv / lambda_beta", j);
        BDF_REQUIRE(!e.feat || e.feat->m == e.N, BDF_ERR_ARG, "bdf_gibbs_create: entity %d has %lld rows but its feature matrix %lld", j,
                    (long long)e.N, e.feat ? (long long)e.feat->m : 0LL);
        for (int t = 0; t < e.n_terms; t++) {
            BDF_REQUIRE(e.terms[t].rel, BDF_ERR_ARG, "bdf_gibbs_create: entity %d term %d has no relation", j, t);
            for (int k = 0; k < e.terms[t].rel->n_modes; k++)
                BDF_REQUIRE(e.terms[t].entity_of_mode[k] >= 0 && e.terms[t].entity_of_mode[k] < n_entities, BDF_ERR_ARG,
                            "bdf_gibbs_create: entity %d term %d mode %d names entity %d", j, t, k, e.terms[t].entity_of_mode[k]);
        }
    }
    BDF_HIP(hipSetDevice(rows_ctx->device));
    bdf_gibbs *g = new bdf_gibbs();
    g->rows = rows_ctx; g->hyper = g->pred = nullptr; g->D = D; g->test = nullptr; g->stats_dev = nullptr;
    g->n_pred = 0; g->comm = nullptr; g->ready_dev = nullptr; g->polling = false;
    g->debug = false; g->host_wait_us = g->host_enqueue_us = 0.0; g->n_sweeps = 0;
    int rc;
    if ((rc = make_side_ctx(rows_ctx, {}, true, &g->hyper)) || (rc = make_side_ctx(rows_ctx, {g->hyper}, false, &g->pred))) { bdf_gibbs_destroy(g); return rc; }
    g->ready_dev = nullptr;
    g->polling = rows_ctx->reserve_cus > 0 && g->hyper->on_reserved && !getenv("BDF_NO_POLL");
    if (g->polling) {
        bool conc = false;
        if ((rc = streams_concurrent(rows_ctx->stream, g->hyper->stream, &conc))) { bdf_gibbs_destroy(g); return rc; }
        g->polling = conc;              // kernels of the two streams do not run side by side here (a profiler serialising them): events
    }
    g->debug = getenv("BDF_DEBUG") != nullptr;
    if (g->debug)
        fprintf(stderr, "[bdf_gibbs] reserve_cus=%d hyperprior stream on reserved CUs=%d polling=%d\n", rows_ctx->reserve_cus,
                g->hyper->on_reserved, (int)g->polling);
    // (every event starts out NULL and bdf_gibbs_destroy skips those: a failure below releases what exists so far)
    struct Guard { bdf_gibbs *g; ~Guard() { if (g) bdf_gibbs_destroy(g); } } guard{g};
    BDF_HIP(hipMalloc((void **)&g->ready_dev, (size_t)n_entities * sizeof(uint32_t)));
    BDF_HIP(hipMemsetAsync(g->ready_dev, 0, (size_t)n_entities * sizeof(uint32_t), rows_ctx->stream));
    BDF_HIP(hipStreamSynchronize(rows_ctx->stream));
    g->ent.resize((size_t)n_entities);
    for (int j = 0; j < n_entities; j++) {
        auto &E = g->ent[(size_t)j];
        E.d = ents[j]; E.cur = 0;
        BDF_HIP(hipEventCreate(&E.ev_rows));          // (they ride on dispatch packets: plain events)
        BDF_HIP(hipEventCreate(&E.ev_hyper));
        if (E.d.feat) BDF_HIP(hipEventCreateWithFlags(&E.ev_beta, hipEventDisableTiming));
    }
    for (int k = 0; k < 3; k++) BDF_HIP(hipEventCreateWithFlags(&g->ev_pred[k], hipEventDisableTiming));
    if (g->polling && !getenv("BDF_NO_COUNTER_HANDOVER")) {
        const size_t bytes = (size_t)BDF_DONE_SHARDS * BDF_DONE_STRIDE * sizeof(uint32_t);
        for (int j = 0; j < n_entities; j++) {
            BDF_HIP(hipMalloc((void **)&g->ent[(size_t)j].done_dev, bytes));
            BDF_HIP(hipMemsetAsync(g->ent[(size_t)j].done_dev, 0, bytes, rows_ctx->stream));
        }
        BDF_HIP(hipStreamSynchronize(rows_ctx->stream));
    }
    guard.g = nullptr;
    *out = g;
    return BDF_OK;
}

extern "C" int bdf_gibbs_destroy(bdf_gibbs *g)
{
    if (!g) return BDF_OK;
    if (g->rows) (void)hipStreamSynchronize(g->rows->stream);
    if (g->debug && g->n_sweeps)
        fprintf(stderr, "[bdf_gibbs] %llu sweeps: host enqueue %.1f us per sweep, host wait for the device %.1f us per sweep\n",
                (unsigned long long)g->n_sweeps, g->host_enqueue_us / (double)g->n_sweeps, g->host_wait_us / (double)g->n_sweeps);
    for (auto &E : g->ent) {
        if (E.ev_rows) (void)hipEventDestroy(E.ev_rows);
        if (E.ev_hyper) (void)hipEventDestroy(E.ev_hyper);
        if (E.ev_beta) (void)hipEventDestroy(E.ev_beta);
        if (E.done_dev) (void)hipFree(E.done_dev);
    }
    for (int k = 0; k < 3; k++)
        if (g->ev_pred[k]) (void)hipEventDestroy(g->ev_pred[k]);
    if (g->ready_dev) (void)hipFree(g->ready_dev);
    if (g->rel_sse) (void)hipFree(g->rel_sse);
    if (g->pred) bdf_ctx_destroy(g->pred);
    if (g->hyper) bdf_ctx_destroy(g->hyper);
    delete g;
    return BDF_OK;
}

extern "C" int bdf_gibbs_contexts(bdf_gibbs *g, bdf_ctx **hyper, bdf_ctx **pred)
{
    BDF_REQUIRE(g, BDF_ERR_ARG, "bdf_gibbs_contexts: NULL argument");
    if (hyper) *hyper = g->hyper;
    if (pred) *pred = g->pred;
    return BDF_OK;
}

extern "C" int bdf_ctx_stream(const bdf_ctx *ctx, void **stream)
{
    BDF_REQUIRE(ctx && stream, BDF_ERR_ARG, "bdf_ctx_stream: NULL argument");
    *stream = (void *)ctx->stream;
    return BDF_OK;
}

extern "C" int bdf_gibbs_set_test(bdf_gibbs *g, bdf_pairs *pairs, const int32_t *entity_of_mode, double mean_value,
                                  double clamp_lo, double clamp_hi, double class_cut, double *stats_dev)
{
    BDF_REQUIRE(g && pairs && entity_of_mode && stats_dev, BDF_ERR_ARG, "bdf_gibbs_set_test: NULL argument");
    for (int k = 0; k < pairs->n_modes; k++) {
        BDF_REQUIRE(entity_of_mode[k] >= 0 && entity_of_mode[k] < (int)g->ent.size(), BDF_ERR_ARG, "bdf_gibbs_set_test: mode %d names entity %d", k, entity_of_mode[k]);
        g->test_entity[k] = entity_of_mode[k];
    }
    g->test = pairs; g->test_mean = mean_value; g->clamp_lo = clamp_lo; g->clamp_hi = clamp_hi; g->class_cut = class_cut;
    g->stats_dev = stats_dev;
    return BDF_OK;
}

extern "C" int bdf_gibbs_set_relations(bdf_gibbs *g, int n_relations, const bdf_gibbs_relation *rels)
{
    BDF_REQUIRE(g && n_relations >= 0 && (n_relations == 0 || rels), BDF_ERR_ARG, "bdf_gibbs_set_relations: bad argument");
    for (int k = 0; k < n_relations; k++) {
        const bdf_gibbs_relation &r = rels[k];
        BDF_REQUIRE(r.rel && r.alpha_dev, BDF_ERR_ARG, "bdf_gibbs_set_relations: relation %d has no handle or no alpha_dev", k);
        BDF_REQUIRE(!(r.alpha_sample || r.feat) || r.train, BDF_ERR_ARG, "bdf_gibbs_set_relations: relation %d needs its observations as pairs (train)", k);
        BDF_REQUIRE(!r.feat || (r.beta && r.linear), BDF_ERR_ARG, "bdf_gibbs_set_relations: relation %d has features but no beta / linear buffer", k);
        BDF_REQUIRE(!r.feat_test || r.test_baseline, BDF_ERR_ARG, "bdf_gibbs_set_relations: relation %d has test features but no baseline buffer", k);
        for (int m = 0; m < r.rel->n_modes; m++)
            BDF_REQUIRE(r.entity_of_mode[m] >= 0 && r.entity_of_mode[m] < (int)g->ent.size(), BDF_ERR_ARG,
                        "bdf_gibbs_set_relations: relation %d mode %d names entity %d", k, m, r.entity_of_mode[m]);
    }
    g->rels.assign(rels, rels + n_relations);
    if (n_relations > 0 && !g->rel_sse) BDF_HIP(hipMalloc((void **)&g->rel_sse, 8 * sizeof(double)));
    return BDF_OK;
}

namespace {
// the registered relation of a term (by its bdf_rel), or NULL
const bdf_gibbs_relation *relation_of(const bdf_gibbs *g, const bdf_rel *rel)
{
    for (const auto &r : g->rels)
        if (r.rel == rel) return &r;
    return nullptr;
}

// macau.jl:83-92 on the row stream, before the entities' rows
int update_relations(bdf_gibbs *g)
{
    bdf_ctx *R = g->rows;
    const int D = g->D;
    int rc;
    for (const auto &r : g->rels) {
        if (!r.alpha_sample && !r.feat) continue;
        const double *fac[BDF_MAX_MODES];
        for (int m = 0; m < r.rel->n_modes; m++) {
            const auto &O = g->ent[(size_t)r.entity_of_mode[m]];
            fac[m] = O.d.sample[O.cur];
        }
        if (r.alpha_sample) {
            // err' err over this rank's block (the pairs carry linear_values as their baseline), summed over the ranks
            if ((rc = bdf_predict_sse(R, r.train, D, fac, r.mean_value, nullptr, g->rel_sse))) return rc;
            if (g->comm && (rc = bdf_sum_ranks(R, g->comm, g->rel_sse + 1, 1))) return rc;
            if ((rc = bdf_sample_alpha(R, r.alpha_lambda0, r.alpha_nu0, r.nnz, g->rel_sse + 1, r.rel_tag, r.alpha_dev))) return rc;
        }
        if (r.feat) {
            if ((rc = bdf_sample_beta_rel_impl(R, g->comm, r.feat, r.train, r.first_obs, D, fac, r.mean_value, 1.0, r.alpha_dev, r.lambda_beta,
                                               r.rel_tag, r.beta, r.linear + r.first_obs, nullptr)))
                return rc;
            if (g->comm) {          // every rank's row kernels read linear_values of their own rows' observations
                if ((rc = bdf_allgather_block(R, g->comm, r.linear, sizeof(double) * (size_t)r.obs_block)) || (rc = bdf_allgather_join(R, g->comm))) return rc;
            }
            if (r.feat_test) {
                // the test pairs' baseline is read by the prediction stream: its updates so far must have completed
                for (int k = 0; k < 3; k++)
                    if (g->n_pred > (uint64_t)k) BDF_HIP(hipStreamWaitEvent(R->stream, g->ev_pred[(g->n_pred - 1 - (uint64_t)k) % 3], 0));
                if ((rc = bdf_feat_linear(R, r.feat_test, r.beta, r.mean_value, r.test_baseline))) return rc;
            }
        }
    }
    return BDF_OK;
}
}  // namespace

extern "C" int bdf_gibbs_set_comm(bdf_gibbs *g, bdf_comm *comm)
{
    BDF_REQUIRE(g, BDF_ERR_ARG, "bdf_gibbs_set_comm: NULL argument");
    g->comm = comm;
    return BDF_OK;
}

// measurement: the rows of one entity and nothing else -- the launch bdf_gibbs_sweep makes for it (same kernel variant, same
// inputs), without the hyperprior update, the exchange or the prediction update.  The chain's state is not kept consistent.
namespace {
// the hyperprior's degrees of freedom: nu0 (+ numF with side information and full_lambda_u, macau.jl:124-129)
double hyper_nu(const bdf_gibbs_entity &e) { return e.nu0 + ((e.feat && e.full_lambda_u) ? (double)e.feat->n : 0.0); }
}  // namespace

extern "C" int bdf_gibbs_rows_only(bdf_gibbs *g, int entity, uint32_t sweep)
{
    BDF_REQUIRE(g && entity >= 0 && entity < (int)g->ent.size(), BDF_ERR_ARG, "bdf_gibbs_rows_only: bad argument");
    bdf_ctx *R = g->rows;
    auto &E = g->ent[(size_t)entity];
    const bdf_gibbs_entity &e = E.d;
    R->sweep_host = sweep;
    bdf_term terms[BDF_MAX_TERMS];
    for (int t = 0; t < e.n_terms; t++) {
        terms[t].rel = e.terms[t].rel; terms[t].mode = e.terms[t].mode; terms[t]._pad = 0;
        terms[t].alpha = e.terms[t].alpha; terms[t].mean_value = e.terms[t].mean_value; terms[t].linear_values = nullptr;
        terms[t].alpha_dev = nullptr;
        if (const bdf_gibbs_relation *gr = relation_of(g, e.terms[t].rel)) {
            terms[t].alpha_dev = gr->alpha_dev;
            terms[t].linear_values = gr->feat ? gr->linear : nullptr;
        }
        for (int k = 0; k < BDF_MAX_MODES; k++) terms[t].factors[k] = nullptr;
        for (int k = 0; k < e.terms[t].rel->n_modes; k++) {
            const auto &O = g->ent[(size_t)e.terms[t].entity_of_mode[k]];
            terms[t].factors[k] = O.d.sample[O.cur];
        }
    }
    const int nxt = (E.cur + 1) % 3;
    const int nch = e.terms[0].rel->chunks;
    int rc;
    for (int c = 0; c < nch; c++)
        if ((rc = bdf_sample_rows(R, g->D, e.N, e.n_terms, terms, e.feat ? e.mu_matrix : e.mu, e.feat ? 1 : 0, e.Lambda, e.tag, c, nch,
                                  e.sample[nxt], (E.hyper_recorded && !e.feat) ? e.prior_pack : nullptr)))
            return rc;
    E.cur = nxt;
    return BDF_OK;
}

// set-up: bring the device to its working state (clocks, power gating: an iteration right after idle time runs ~10 % slower than
// the same iteration 30 ms into sustained work, and what ran before matters as much as how long: DESIGN.md section 6) with FULL
// iterations whose results are discarded -- rows of every entity, exchanges, hyperprior chains, beta, the relation models, the
// prediction kernel without running state -- under iteration numbers no real iteration uses.  The chain's state (every entity's
// current sample, (mu, Lambda), sums, prior pack, draws, beta, uhat, per-row prior means, lambda_beta, the relations' alpha, beta
// and linear_values, the prediction statistics, the "a draw / beta of an earlier iteration exists" flags) is saved first and put
// back bit for bit afterwards; the buffers have rotated (bdf_gibbs_current).  One rank: iterations for `milliseconds`.  Several
// ranks: every rank must make the same number of exchanges -- the first batch of eight is timed, every rank derives a count from
// its own time, and the ranks agree on the mean of the counts (bdf_sum_ranks).
extern "C" int bdf_gibbs_warm_device(bdf_gibbs *g, double milliseconds)
{
    BDF_REQUIRE(g && milliseconds >= 0.0 && milliseconds <= 10000.0, BDF_ERR_ARG, "bdf_gibbs_warm_device: bad argument");
    if (milliseconds <= 0.0) return BDF_OK;
    bdf_ctx *R = g->rows;
    const int D = g->D, n = (int)g->ent.size();
    int rc;
    if ((rc = bdf_gibbs_sync(g))) return rc;
    // the chain's state: every entity's current sample, (mu, Lambda), sums, posterior parameters, prior pack, draws, and with
    // side information beta, uhat, the per-row prior means, Tinv, lambda_beta, the iteration counts
    struct Piece { void *p; size_t bytes; bool sample; int ent; };
    std::vector<Piece> pieces;
    std::vector<char> flags;
    const size_t DD = (size_t)D * D * sizeof(double), Dv = (size_t)D * sizeof(double);
    for (int j = 0; j < n; j++) {
        auto &E = g->ent[(size_t)j];
        const bdf_gibbs_entity &e = E.d;
        pieces.push_back({e.sample[E.cur], (size_t)e.N * Dv, true, j});
        pieces.push_back({e.mu, Dv, false, j}); pieces.push_back({e.Lambda, DD, false, j});
        pieces.push_back({e.sumU, Dv, false, j}); pieces.push_back({e.UUt, DD, false, j});
        if (e.params) pieces.push_back({e.params, Dv + DD, false, j});
        pieces.push_back({e.prior_pack, (size_t)bdf_prior_pack_doubles(D) * sizeof(double), false, j});
        pieces.push_back({e.draws, DD + Dv, false, j});
        if (e.feat) {
            pieces.push_back({e.beta, (size_t)e.feat->n * Dv, false, j});
            pieces.push_back({e.uhat, (size_t)e.N * Dv, false, j});
            pieces.push_back({e.mu_matrix, (size_t)e.N * Dv, false, j});
            pieces.push_back({e.Tinv, DD, false, j});
            pieces.push_back({e.lambda_beta, sizeof(double), false, j});
            if (e.cg_iters) pieces.push_back({e.cg_iters, (size_t)D * sizeof(int32_t), false, j});
        }
        flags.push_back(E.hyper_recorded ? 1 : 0); flags.push_back(E.beta_recorded ? 1 : 0);
    }
    if (g->stats_dev) pieces.push_back({g->stats_dev, 4 * sizeof(double), false, 0});       // the prediction statistics of the last real update
    for (const auto &r : g->rels) {         // the relation models: alpha, relation-level beta, linear_values, the test pairs' baseline
        pieces.push_back({r.alpha_dev, sizeof(double), false, 0});
        if (r.feat) {
            int world = 1, rk = 0;
            if (g->comm) (void)bdf_comm_size(g->comm, &rk, &world);
            pieces.push_back({r.beta, (size_t)r.feat->n * sizeof(double), false, 0});
            pieces.push_back({r.linear, (size_t)r.obs_block * (size_t)world * sizeof(double), false, 0});
            if (r.feat_test) pieces.push_back({r.test_baseline, (size_t)r.feat_test->m * sizeof(double), false, 0});
        }
    }
    size_t total = 0;
    for (auto &pc : pieces) total += (pc.bytes + 255) & ~(size_t)255;
    char *snap = nullptr;
    BDF_HIP(hipMalloc((void **)&snap, std::max<size_t>(total, 256)));
    struct Free { char *p; ~Free() { if (p) (void)hipFree(p); } } guard{snap};
    size_t off = 0;
    for (auto &pc : pieces) {
        BDF_HIP(hipMemcpyAsync(snap + off, pc.p, pc.bytes, hipMemcpyDeviceToDevice, R->stream));
        off += (pc.bytes + 255) & ~(size_t)255;
    }
    BDF_HIP(hipStreamSynchronize(R->stream));
    // full iterations with numbers no real iteration uses; the prediction kernel without running state.  One rank: for the
    // time asked for; several ranks: a fixed count (every rank must make the same number of exchanges)
    const uint32_t keep = R->sweep_host;
    int64_t fixed = 0;                  // several ranks: the iteration count they agreed on (0: not yet known / one rank)
    const auto t0 = std::chrono::steady_clock::now();
    int64_t k = 0;
    rc = BDF_OK;
    while (!rc) {
        for (int b = 0; b < 8 && !rc; b++, k++) rc = bdf_gibbs_sweep(g, 0xfffe0000u + (uint32_t)(k & 0xffff), g->test ? 3 : -1);
        if (rc) break;
        if (g->comm && fixed == 0) {
            // the first batch, timed to its end on this rank; the ranks' counts summed in rank order: the same number everywhere
            if ((rc = bdf_gibbs_sync(g))) break;
            const double per_iter = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 8.0;
            const double mine = std::max(8.0, std::min(20000.0, milliseconds / std::max(per_iter, 1e-3)));
            int rank = 0, world = 1;
            if ((rc = bdf_comm_size(g->comm, &rank, &world))) break;
            void *sv;
            if ((rc = bdf_scratch(R, ((size_t)world + 1) * sizeof(double), &sv))) break;
            double *cnt = (double *)sv;
            BDF_HIP(hipMemcpyAsync(cnt, &mine, sizeof(double), hipMemcpyHostToDevice, R->stream));
            if ((rc = bdf_sum_ranks_into(R, g->comm, cnt, 1, cnt + 1))) break;
            double sum = 0.0;
            BDF_HIP(hipMemcpyAsync(&sum, cnt, sizeof(double), hipMemcpyDeviceToHost, R->stream));
            BDF_HIP(hipStreamSynchronize(R->stream));
            fixed = std::max<int64_t>(8, (int64_t)(sum / (double)world + 0.5));
        }
        if (fixed ? k >= fixed : std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() >= milliseconds) break;
    }
    const int rc_sync = bdf_gibbs_sync(g);
    if (!rc) rc = rc_sync;
    // put the state back: the saved sample into whichever buffer is current now, everything else where it was
    off = 0;
    for (auto &pc : pieces) {
        void *dst = pc.sample ? (void *)g->ent[(size_t)pc.ent].d.sample[g->ent[(size_t)pc.ent].cur] : pc.p;
        BDF_HIP(hipMemcpyAsync(dst, snap + off, pc.bytes, hipMemcpyDeviceToDevice, R->stream));
        off += (pc.bytes + 255) & ~(size_t)255;
    }
    for (int j = 0; j < n; j++) {
        g->ent[(size_t)j].hyper_recorded = flags[(size_t)2 * j] != 0;
        g->ent[(size_t)j].beta_recorded = flags[(size_t)2 * j + 1] != 0;
    }
    BDF_HIP(hipStreamSynchronize(R->stream));
    R->sweep_host = g->hyper->sweep_host = g->pred->sweep_host = keep;
    return rc;
}

// (set-up) whether the entity's hyperprior draw / beta of an earlier iteration exist -- what the next iteration's row launch
// takes its prior pack from and waits for; a caller that runs iterations and then puts the chain's state back (the engine's
// device warm-up) puts these back with it
extern "C" int bdf_gibbs_recorded(const bdf_gibbs *g, int entity, int *hyper, int *beta)
{
    BDF_REQUIRE(g && entity >= 0 && entity < (int)g->ent.size() && hyper && beta, BDF_ERR_ARG, "bdf_gibbs_recorded: bad argument");
    *hyper = g->ent[(size_t)entity].hyper_recorded ? 1 : 0;
    *beta = g->ent[(size_t)entity].beta_recorded ? 1 : 0;
    return BDF_OK;
}

extern "C" int bdf_gibbs_set_recorded(bdf_gibbs *g, int entity, int hyper, int beta)
{
    BDF_REQUIRE(g && entity >= 0 && entity < (int)g->ent.size(), BDF_ERR_ARG, "bdf_gibbs_set_recorded: bad argument");
    g->ent[(size_t)entity].hyper_recorded = hyper != 0;
    g->ent[(size_t)entity].beta_recorded = beta != 0;
    return BDF_OK;
}

extern "C" int bdf_gibbs_time_rows(bdf_gibbs *g, int entity, void *start, void *stop)
{
    BDF_REQUIRE(g && entity >= 0 && entity < (int)g->ent.size(), BDF_ERR_ARG, "bdf_gibbs_time_rows: bad entity");
    g->ent[(size_t)entity].t_start = (hipEvent_t)start;
    g->ent[(size_t)entity].t_stop = (hipEvent_t)stop;
    return BDF_OK;
}

extern "C" int bdf_gibbs_span_rows(bdf_gibbs *g, int entity, void *slot_dev)
{
    BDF_REQUIRE(g && entity >= 0 && entity < (int)g->ent.size(), BDF_ERR_ARG, "bdf_gibbs_span_rows: bad entity");
    g->ent[(size_t)entity].span = (unsigned long long *)slot_dev;
    return BDF_OK;
}

extern "C" int bdf_gibbs_current(const bdf_gibbs *g, int entity, int *buffer)
{
    BDF_REQUIRE(g && buffer && entity >= 0 && entity < (int)g->ent.size(), BDF_ERR_ARG, "bdf_gibbs_current: bad argument");
    *buffer = g->ent[(size_t)entity].cur;
    return BDF_OK;
}

extern "C" int bdf_gibbs_sweep(bdf_gibbs *g, uint32_t sweep, int predict_phase)
{
    BDF_REQUIRE(g, BDF_ERR_ARG, "bdf_gibbs_sweep: NULL argument");
    bdf_ctx *R = g->rows, *H = g->hyper, *P = g->pred;
    const int D = g->D, n = (int)g->ent.size();
    int rc;
    R->sweep_host = H->sweep_host = P->sweep_host = sweep;
    // The row kernels of the NEXT sweep overwrite the buffers that held the rows of sweep - 2, which the prediction update of
    // sweep - 2 reads.  The HOST waits here until an update of some sweeps ago has completed (normally it has, long ago): the
    // device then needs no wait for the prediction stream anywhere, and the host never runs more than a few prediction updates
    // ahead.  (A device-side wait would let row kernels that poll for their prior fill the chip while the prediction kernel
    // they transitively wait for still needs slots for its last workgroups.)
    // (Whether or not THIS iteration has a prediction update: an update enqueued two or more iterations ago must have
    // completed -- iterations without one rotate the buffers all the same.)
    // How many iterations back: 3 (BDF_PRED_LAG, 1..3) -- the most the three buffers allow: iteration s overwrites the rows of
    // iteration s - 3, which the prediction update of s - 3 read.  (Until round 3: 2.  On a quiet host there is no difference;
    // with 3 an enqueue that takes 50 us instead of 23, or a late wake-up, no longer leaves the row stream dry.)
    static const int pred_lag = getenv("BDF_PRED_LAG") ? std::max(1, std::min(3, atoi(getenv("BDF_PRED_LAG")))) : 3;
    const auto t_in = std::chrono::steady_clock::now();
    for (uint64_t back = 1; back <= std::min<uint64_t>(g->n_pred, 3); back++) {
        const uint64_t k = (g->n_pred - back) % 3;
        if (g->pred_at[k] + (uint64_t)pred_lag <= g->n_iter) {          // (and with it the earlier ones)
            // a short spin on the event before the blocking wait: a thread put to sleep here wakes 10-40 us after the event,
            // and with the host one iteration ahead of the device at that moment a late wake-up plus a slow enqueue (50 us on
            // a busy host) leaves the row stream dry
            hipError_t st = hipErrorNotReady;
            const auto t_spin = std::chrono::steady_clock::now();
            while ((st = hipEventQuery(g->ev_pred[k])) == hipErrorNotReady &&
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_spin).count() < 400.0) { }
            if (st == hipErrorNotReady) st = hipEventSynchronize(g->ev_pred[k]);
            BDF_HIP(st);
            break;
        }
    }
    const auto t_go = std::chrono::steady_clock::now();
    // the data-independent part of every entity's hyperprior draw (Bartlett matrix, mean normals): inside the entity's chain
    // launch, beside its partial sums, when every entity is small enough for the one-launch chain (the hyperprior stream is
    // busy ~85 of an iteration's 90 us: a launch of its own at the head of the iteration is 7 us of that); else ahead of the rows
    static const bool fuse_sums_ = !(getenv("BDF_HYPER_FUSE") && atoi(getenv("BDF_HYPER_FUSE")) == 0);
    static const bool one_launch_ = !(getenv("BDF_HYPER_CHAIN") && atoi(getenv("BDF_HYPER_CHAIN")) == 0);
    static const bool draws_ahead_ = getenv("BDF_DRAWS_AHEAD") != nullptr;
    bool draws_in_chain = fuse_sums_ && one_launch_ && !draws_ahead_ && !g->comm;      // (several ranks: the sums are not the chain's, bdf_hyper_sums_ranks)
    for (int j = 0; j < n; j++) draws_in_chain = draws_in_chain && g->ent[(size_t)j].d.N <= 16384;
    if (!draws_in_chain && n <= BDF_DRAWS_BATCH) {
        int64_t Ns[BDF_DRAWS_BATCH]; double nus[BDF_DRAWS_BATCH]; uint32_t tags[BDF_DRAWS_BATCH]; double *outs[BDF_DRAWS_BATCH];
        for (int j = 0; j < n; j++) {
            const bdf_gibbs_entity &e = g->ent[(size_t)j].d;
            Ns[j] = e.n_real; nus[j] = hyper_nu(e); tags[j] = e.tag; outs[j] = e.draws;
        }
        if ((rc = bdf_hyper_draws_batch(H, D, n, Ns, nus, tags, outs))) return rc;
    }
    // the relation models (alpha, relation-level beta and linear_values) with the rows of the previous iteration (macau.jl:83-92)
    if (!g->rels.empty() && (rc = update_relations(g))) return rc;
    for (int j = 0; j < n; j++) {
        auto &E = g->ent[(size_t)j];
        const bdf_gibbs_entity &e = E.d;
        // (mu, Lambda) of the previous iteration: an event wait, or -- draws on reserved CUs -- the row kernel polls for it
        // (with side information the prior mean is a matrix, computed here from mu: nothing to poll for)
        // (several ranks: event waits unless BDF_POLL_WITH_COMM is set -- the draws of an iteration depend on nothing the rows of
        // the next one produce, so polling cannot deadlock with the exchange's kernels either, but that schedule has only run
        // with a one-rank RCCL communicator: tools/soak_determinism.py rccl)
        static const bool poll_with_comm = getenv("BDF_POLL_WITH_COMM") != nullptr;
        const bool poll = g->polling && (!g->comm || poll_with_comm) && E.hyper_recorded && !e.feat;
        if (E.hyper_recorded && !poll) BDF_HIP(hipStreamWaitEvent(R->stream, E.ev_hyper, 0));
        // side information: uhat = (F beta)' with the beta of the previous iteration, per-row prior means mu + uhat (macau.jl:103-104)
        if (e.feat && (rc = bdf_uhat(R, e.feat, D, e.beta, e.mu, e.uhat, e.mu_matrix))) return rc;
        // the data-independent part of the hyperprior draw (Bartlett matrix, mean normals): beside the rows -- for all entities
        // in one launch at the head of the iteration when they are few
        if (!draws_in_chain && n > BDF_DRAWS_BATCH && (rc = bdf_hyper_draws(H, D, e.n_real, hyper_nu(e), e.tag, e.draws))) return rc;
        bdf_term terms[BDF_MAX_TERMS];
        for (int t = 0; t < e.n_terms; t++) {
            terms[t].rel = e.terms[t].rel; terms[t].mode = e.terms[t].mode; terms[t]._pad = 0;
            terms[t].alpha = e.terms[t].alpha; terms[t].mean_value = e.terms[t].mean_value; terms[t].linear_values = nullptr;
            terms[t].alpha_dev = nullptr;
            if (const bdf_gibbs_relation *gr = relation_of(g, e.terms[t].rel)) {        // a relation with a model of its own
                terms[t].alpha_dev = gr->alpha_dev;
                terms[t].linear_values = gr->feat ? gr->linear : nullptr;
            }
            for (int k = 0; k < BDF_MAX_MODES; k++) terms[t].factors[k] = nullptr;
            for (int k = 0; k < e.terms[t].rel->n_modes; k++) {
                const auto &O = g->ent[(size_t)e.terms[t].entity_of_mode[k]];
                terms[t].factors[k] = O.d.sample[O.cur];
            }
        }
        const int nxt = (E.cur + 1) % 3;
        // the completion event rides on the row kernel's dispatch (a caller-supplied timing pair takes its place)
        hipEvent_t done = E.t_stop ? E.t_stop : E.ev_rows;
        const int nch = e.terms[0].rel->chunks;                    // 1 unless the relations were created with a layout
        // The hand-over to the entity's hyperprior chain.  By default an event on the row kernel's dispatch and a wait on the
        // hyperprior stream -- which costs that stream ~10 us per chain even when the event completed long before, and the row
        // stream ~1.7 us per launch.  When the chain is the one-launch kind and runs on reserved CUs (the schedule in which the
        // row kernels poll for the draw), the chain is enqueued AT ONCE instead and its partial-sum workgroups poll a counter
        // the row waves add to when their rows are in memory (SampleArgs::done; k_rows_col only -- a launch that takes another
        // kernel reports -1 and gets the event).  No cycle: the rows of iteration t poll for the draw of t - 1, enqueued before.
        const bool counter = E.done_dev && draws_in_chain && g->polling && !g->comm && !e.feat && nch == 1 && !E.t_stop;
        const bool pred_waits = j == n - 1 && g->test && predict_phase >= 0;
        bool by_counter = false;
        for (int c = 0; c < nch; c++) {
            R->time_start = (c == 0) ? E.t_start : nullptr;
            R->rows_span = E.span;
            R->time_stop = (c == nch - 1 && !g->comm && !(counter && !pred_waits)) ? done : nullptr;
            if (poll) { R->rows_ready = g->ready_dev + j; R->rows_ready_want = E.epoch; }
            if (counter) R->rows_done = E.done_dev;
            if ((rc = bdf_sample_rows(R, D, e.N, e.n_terms, terms, e.feat ? e.mu_matrix : e.mu, e.feat ? 1 : 0, e.Lambda, e.tag, c, nch,
                                      e.sample[nxt], (E.hyper_recorded && !e.feat) ? e.prior_pack : nullptr)))
                return rc;
            if (counter && R->rows_done_added >= 0) { by_counter = true; E.done_target += (uint32_t)R->rows_done_added; }
            if (g->comm && (rc = bdf_allgather_rows(R, g->comm, D, e.N, e.sample[nxt], c, nch))) return rc;
        }
        if (g->comm) {
            if ((rc = bdf_allgather_join(R, g->comm))) return rc;       // the row stream continues after the last chunk's exchange
            BDF_HIP(hipEventRecord(done, R->stream));
        } else if (counter && !pred_waits && !by_counter) {
            BDF_HIP(hipEventRecord(done, R->stream));                   // (the launch took another kernel: the event after all)
        }
        E.t_start = E.t_stop = nullptr;
        E.span = nullptr;
        E.cur = nxt;
        if (by_counter) { H->hyper_wait = E.done_dev; H->hyper_wait_target = E.done_target; }
        else BDF_HIP(hipStreamWaitEvent(H->stream, done, 0));
        static const bool fuse_sums = !(getenv("BDF_HYPER_FUSE") && atoi(getenv("BDF_HYPER_FUSE")) == 0);
        H->hyper_fuse = fuse_sums;        // small entities: the draw adds the sums' partials itself (one launch fewer)
        // side information: U = sample - uhat, T^-1 = WI + beta' beta lambda_beta with the beta of the previous iteration
        const double *Tinv = e.WI;
        if (e.feat && e.full_lambda_u) {
            if (E.beta_recorded) BDF_HIP(hipStreamWaitEvent(H->stream, E.ev_beta, 0));
            if ((rc = bdf_hyper_feature_terms(H, D, e.feat->n, e.beta, e.WI, e.lambda_beta, e.Tinv))) return rc;
            Tinv = e.Tinv;
        }
        // (several ranks: every rank sums its own rows, the D + D^2 partial sums are added over the ranks in rank order)
        if ((rc = g->comm ? bdf_hyper_sums_ranks(H, g->comm, D, e.N, nch, e.sample[E.cur], e.feat ? e.uhat : nullptr, e.sumU, e.UUt)
                          : bdf_hyper_sums(H, D, e.N, e.sample[E.cur], e.feat ? e.uhat : nullptr, e.sumU, e.UUt)))
            return rc;
        H->hyper_chain_draws = draws_in_chain ? e.draws : nullptr;
        H->time_h_stop = E.ev_hyper;
        H->hyper_ready = g->ready_dev + j;
        H->hyper_ready_value = ++E.epoch;
        if ((rc = bdf_hyper_sample(H, D, e.n_real, e.sumU, e.UUt, e.mu0, e.b0, Tinv, hyper_nu(e), e.tag, e.mu, e.Lambda, e.params, e.prior_pack, e.draws)))
            return rc;
        E.hyper_recorded = true;
        if (j == n - 1 && g->test && predict_phase >= 0) BDF_HIP(hipStreamWaitEvent(P->stream, done, 0));
    }
    // side information: beta of every entity that has it, from this iteration's rows and (mu, Lambda) (macau.jl:138-140)
    for (int j = 0; j < n; j++) {
        auto &E = g->ent[(size_t)j];
        const bdf_gibbs_entity &e = E.d;
        if (!e.feat) continue;
        BDF_HIP(hipStreamWaitEvent(R->stream, E.ev_hyper, 0));
        if ((rc = bdf_sample_beta_ranks(R, g->comm, e.feat, D, e.sample[E.cur], e.mu, e.Lambda, e.lambda_beta, e.use_ff, e.tol, 0,
                                        e.sample_lambda_beta, e.lb_nu, e.lb_mu, e.tag, e.beta, nullptr, e.cg_iters)))
            return rc;
        BDF_HIP(hipEventRecord(E.ev_beta, R->stream));
        E.beta_recorded = true;
    }
    if (g->test && predict_phase >= 0) {
        const double *fac[BDF_MAX_MODES];
        for (int k = 0; k < g->test->n_modes; k++) {
            const auto &O = g->ent[(size_t)g->test_entity[k]];
            fac[k] = O.d.sample[O.cur];
        }
        // (phase 3, set-up only: the same kernel on the same pairs, statistics of this sample into stats_dev, no running state)
        if ((rc = predict_phase == 3 ? bdf_predict_sse(P, g->test, D, fac, g->test_mean, nullptr, g->stats_dev)
                                     : bdf_predict_update(P, g->test, D, fac, g->test_mean, predict_phase, g->clamp_lo, g->clamp_hi, g->class_cut, g->stats_dev)))
            return rc;
        BDF_HIP(hipEventRecord(g->ev_pred[g->n_pred % 3], P->stream));
        g->pred_at[g->n_pred % 3] = g->n_iter;
        g->n_pred++;
    }
    g->n_iter++;
    if (g->debug) {
        g->host_wait_us += std::chrono::duration<double, std::micro>(t_go - t_in).count();
        g->host_enqueue_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_go).count();
        g->n_sweeps++;
    }
    return BDF_OK;
}

extern "C" int bdf_gibbs_sync(bdf_gibbs *g)
{
    BDF_REQUIRE(g, BDF_ERR_ARG, "bdf_gibbs_sync: NULL argument");
    int rc;
    if ((rc = bdf_ctx_sync(g->pred)) || (rc = bdf_ctx_sync(g->hyper))) return rc;
    return bdf_ctx_sync(g->rows);
}
