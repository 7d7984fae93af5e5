// bdf_comm.hip -- the exchange step of the multi-GPU sweep: after a rank has sampled its rows of an entity, every rank needs
// every row before the next entity's rows are sampled (the reference ships the whole factor to every worker per call,
// src/sampling.jl:155-167).  With the row layout of bdf_layout_build a chunk of the factor matrix is one contiguous,
// rank-major region, so the exchange is an IN-PLACE all-gather -- no packing kernels -- on its own stream, and the row
// kernel of chunk c + 1 runs while chunk c is exchanged.
//
// Transport: RCCL (ncclAllGather over xGMI), resolved with dlopen at the first use so that the library loads on hosts
// without RCCL; or a host callback (test rigs with several ranks on one GPU, where RCCL refuses to run: the block is staged
// through host memory and the caller's function -- e.g. a gloo all-gather -- does the exchange).
//
// LARGE exchanges -- bdf_comm_enable_peer: DIRECT ALL-PAIRS COPIES.  RCCL's all-gather is a ring: every byte crosses P - 1
// links one after the other, and a ring uses one of a GPU's seven xGMI links at a time -- configuration C4's 5.12 GB user
// factor (640 MB per rank on 8 GPUs) takes ~29 ms that way.  MI355X's xGMI is point-to-point, all pairs
// connected: every rank can PULL its P - 1 missing blocks from their owners at once, one copy per link: 640 MB per link,
// ~4.2 ms.  The ranks are processes: each exports the allocation its block lives in (hipIpcGetMemHandle), opens its peers'
// (hipIpcOpenMemHandle, cached per handle), and an exchange is P - 1 concurrent device-to-device copies on P - 1 streams.
// What orders them is THE DEVICE (round 6; until then the host: a stream synchronisation on either side of every exchange):
//   * every rank owns a ring of INTERPROCESS events (hipEventInterprocess; the peers open their handles once).  Exchange k: the
//     owner records ready[k % R] on the row stream behind the row kernel that wrote its block; each peer's copy stream for that
//     link waits for the opened event (a barrier packet for the command processor: 3-9 us of host time, no CU, no host wait --
//     tools/ipc_event_probe.hip) and then copies;
//   * the host's all-gather still carries the control message (memory handle, offset, size, an error code the ranks agree on),
//     but it orders only the CALLS -- a wait captures the latest record made before it, so every rank must have recorded before
//     any rank waits -- not the device: nobody synchronises a stream.  The host runs ahead of the device; the row kernel of
//     chunk c + 1 is enqueued, and runs, while chunk c's copies are in flight (bdf_gibbs_sweep), and the row stream waits for
//     the copies only in bdf_allgather_join, before the next reader of the factor;
//   * an owner rewrites a block three iterations later (the entity's three sample buffers rotate); by then every peer's copy of
//     it has completed: the owner's row kernel is stream-ordered behind its joins of the two iterations between, which waited
//     for the peers' ready events of those iterations, which the peers recorded behind THEIR joins of the iteration before.
//     That argument needs rotating buffers: only bdf_allgather_rows takes this path (bdf_allgather_block's callers reuse one
//     buffer call after call -- ADVICE round 5 -- and keep the communicator's own transport).
// OFF unless the caller turns it on (BDF_COMM_PEER=1 in the Python host): UNMEASURED on several GPUs -- the pool has one-GPU
// boxes -- and exercised by two processes on one GPU (tests/test_gpu_macau.py, profiles/r06_peer_overlap_timeline.txt).
#include "bdf_common.h"
#include <dlfcn.h>
#include <unistd.h>
#include <random>

namespace {
struct NcclId { char internal[128]; };
typedef int (*fn_get_id)(NcclId *);
typedef int (*fn_init_rank)(void **, int, NcclId, int);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_err)(int);
struct Rccl {
    void *h = nullptr;
    fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_all_gather all_gather = nullptr; fn_destroy destroy = nullptr;
    fn_err err = nullptr;
};
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.h) return BDF_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    BDF_REQUIRE(h, BDF_ERR_HIP, "bdf_comm: librccl.so not found (%s)", dlerror());
    g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.err = (fn_err)dlsym(h, "ncclGetErrorString");
    BDF_REQUIRE(g_rccl.get_id && g_rccl.init_rank && g_rccl.all_gather && g_rccl.destroy, BDF_ERR_HIP, "bdf_comm: librccl.so lacks a symbol");
    g_rccl.h = h;
    return BDF_OK;
}
#define BDF_NCCL(expr)                                                                                        \
    do {                                                                                                      \
        int e__ = (expr);                                                                                     \
        if (e__ != 0) { bdf_set_error("%s failed: %s", #expr, g_rccl.err ? g_rccl.err(e__) : "?"); return BDF_ERR_HIP; } \
    } while (0)
}  // namespace

struct bdf_comm {
    bdf_ctx *ctx;
    int rank, world;
    void *nccl;                      // ncclComm_t, or NULL with the host transport
    bdf_exchange_fn cb;
    void *cb_user;
    hipStream_t stream;              // the exchange runs here
    hipEvent_t ev_rows, ev_done;
    std::vector<char> hsend, hrecv;
    // direct all-pairs copies for large exchanges (bdf_comm_enable_peer)
    bdf_exchange_fn peer_cb = nullptr;
    void *peer_user = nullptr;
    size_t peer_min_bytes = 0;
    std::vector<hipStream_t> pstreams;                       // one per peer
    std::vector<hipEvent_t> pdone;                           // ... and the event its latest copy records
    std::vector<std::pair<std::vector<char>, void *>> opened; // (peer rank byte + handle bytes) -> mapped base
    int64_t peer_exchanges = 0, peer_bytes = 0;
    // the device-side order of the copies: a ring of interprocess events per rank ("my block of exchange k is complete"), the
    // peers' opened once (set up by the first exchange: a collective)
    static constexpr int RING = 4;
    hipEvent_t ready[RING] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> peer_ready;                      // [peer rank * RING + slot]; the own rank's entries stay NULL
    bool ring_open = false;
    uint64_t seq = 0;                                        // exchanges by peer copies so far (the same on every rank)
    bool join_pending = false;                               // copies enqueued since the last bdf_allgather_join
    uint64_t nonce = 0;                                      // identifies this process among the ranks
};

namespace {
struct PeerMsg {
    hipIpcMemHandle_t handle;       // the allocation the block lives in
    uint64_t offset;                // of the exchanged region's start inside it
    uint64_t bytes;                 // per rank (checked: every rank exchanges the same amount)
    uint64_t nonce;                 // of the sending process (two ranks in ONE process cannot open each other's handles: refused)
    uint64_t seq;                   // the sender's count of peer exchanges (checked: the ranks move in step)
    int32_t err, _pad;              // the sender's failure before the message, if any: every rank returns an error then
};
struct RingMsg {
    hipIpcEventHandle_t ready[bdf_comm::RING];
    uint64_t nonce;
    int32_t err, _pad;
};

// (collective) the ring of interprocess events: created, exported, the peers' opened
int peer_open_ring(bdf_comm *c)
{
    RingMsg mine;
    memset(&mine, 0, sizeof(mine));
    mine.nonce = c->nonce;
    hipError_t e = hipSuccess;
    for (int k = 0; k < bdf_comm::RING && e == hipSuccess; k++) {
        if (!c->ready[k]) e = hipEventCreateWithFlags(&c->ready[k], hipEventDisableTiming | hipEventInterprocess);
        if (e == hipSuccess) e = hipIpcGetEventHandle(&mine.ready[k], c->ready[k]);
    }
    mine.err = e == hipSuccess ? 0 : 1;
    std::vector<RingMsg> all((size_t)c->world);
    const int rc = c->peer_cb(c->peer_user, &mine, all.data(), sizeof(RingMsg));
    BDF_REQUIRE(rc == 0, BDF_ERR_HIP, "bdf_comm (peer copies): the host exchange function returned %d", rc);
    for (int p = 0; p < c->world; p++)
        BDF_REQUIRE(all[(size_t)p].err == 0, BDF_ERR_HIP, "bdf_comm (peer copies): rank %d could not create or export its interprocess events%s%s", p,
                    p == c->rank ? ": " : "", p == c->rank ? hipGetErrorString(e) : "");
    c->peer_ready.assign((size_t)c->world * bdf_comm::RING, nullptr);
    bool ok = true;
    for (int p = 0; p < c->world && ok; p++) {
        if (p == c->rank) continue;
        if (all[(size_t)p].nonce == c->nonce) { ok = false; break; }       // a peer in this very process
        for (int k = 0; k < bdf_comm::RING && ok; k++)
            ok = hipIpcOpenEventHandle(&c->peer_ready[(size_t)p * bdf_comm::RING + k], all[(size_t)p].ready[k]) == hipSuccess;
    }
    // (agreed on: a rank that could not open a handle must not leave the others inside the next collective)
    int32_t mine_ok = ok ? 0 : 1;
    std::vector<int32_t> oks((size_t)c->world);
    const int rc2 = c->peer_cb(c->peer_user, &mine_ok, oks.data(), sizeof(int32_t));
    BDF_REQUIRE(rc2 == 0, BDF_ERR_HIP, "bdf_comm (peer copies): the host exchange function returned %d", rc2);
    for (int p = 0; p < c->world; p++)
        BDF_REQUIRE(oks[(size_t)p] == 0, BDF_ERR_HIP, "bdf_comm (peer copies): rank %d could not open its peers' interprocess events", p);
    c->ring_open = true;
    return BDF_OK;
}

// in-place all-gather of `bytes` per rank at region + r * bytes by direct copies from the owners' mappings, ordered on the device:
// this rank's block is complete when the work enqueued so far on ctx's stream is; the copies land behind c->stream (bdf_allgather_join)
int peer_allgather(bdf_ctx *ctx, bdf_comm *c, char *region, size_t bytes)
{
    int rc;
    if (!c->ring_open && (rc = peer_open_ring(c))) return rc;
    const int slot = (int)(c->seq % bdf_comm::RING);
    PeerMsg mine;
    memset(&mine, 0, sizeof(mine));
    void *base = nullptr;
    size_t asize = 0;
    hipError_t e = hipMemGetAddressRange((hipDeviceptr_t *)&base, &asize, (hipDeviceptr_t)region);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&mine.handle, base);
    // "my block of this exchange is complete": behind everything enqueued on the row stream so far.  (Re-recorded RING exchanges
    // later: by then every peer has made its wait for this one -- it did so right behind this exchange's control message, and the
    // control messages of the exchanges between are collectives.)
    if (e == hipSuccess) e = hipEventRecord(c->ready[slot], ctx->stream);
    mine.offset = (uint64_t)(region - (char *)base);
    mine.bytes = (uint64_t)bytes;
    mine.nonce = c->nonce;
    mine.seq = c->seq;
    mine.err = e == hipSuccess ? 0 : 1;
    std::vector<PeerMsg> all((size_t)c->world);
    rc = c->peer_cb(c->peer_user, &mine, all.data(), sizeof(PeerMsg));      // the control messages; orders the CALLS (record before wait), not the device
    BDF_REQUIRE(rc == 0, BDF_ERR_HIP, "bdf_comm (peer copies): the host exchange function returned %d", rc);
    for (int p = 0; p < c->world; p++) {
        const PeerMsg &m = all[(size_t)p];
        BDF_REQUIRE(m.err == 0, BDF_ERR_HIP, "bdf_comm (peer copies): rank %d could not export its block%s%s", p, p == c->rank ? ": " : "",
                    p == c->rank ? hipGetErrorString(e) : "");
        BDF_REQUIRE(m.bytes == (uint64_t)bytes && m.seq == c->seq, BDF_ERR_ARG, "bdf_comm (peer copies): rank %d is at exchange %llu of %llu bytes, this rank at %llu of %llu",
                    p, (unsigned long long)m.seq, (unsigned long long)m.bytes, (unsigned long long)c->seq, (unsigned long long)bytes);
    }
    // the peers' mappings (cached per handle).  Every rank sees every exchange, so whether this one brings a handle that is new is
    // the same everywhere: then -- and only then -- the ranks agree on the outcome of opening it before anyone copies (a rank that
    // failed alone would leave the others inside the next collective)
    std::vector<const char *> srcs((size_t)c->world, nullptr);
    bool any_new = false, opened_ok = true;
    for (int p = 0; p < c->world; p++) {
        const PeerMsg &m = all[(size_t)p];
        std::vector<char> key(1 + sizeof(hipIpcMemHandle_t));
        key[0] = (char)p;
        memcpy(key.data() + 1, &m.handle, sizeof(hipIpcMemHandle_t));
        void *pbase = nullptr;
        bool seen = false;
        for (auto &o : c->opened)
            if (o.first == key) { pbase = o.second; seen = true; break; }
        if (!seen) {
            any_new = true;
            if (p != c->rank) {
                if (m.nonce == c->nonce) { opened_ok = false; bdf_set_error("bdf_comm (peer copies): rank %d lives in this process", p); }
                else if (hipIpcOpenMemHandle(&pbase, m.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                    opened_ok = false; pbase = nullptr;
                    bdf_set_error("bdf_comm (peer copies): hipIpcOpenMemHandle of rank %d's allocation failed", p);
                }
            }
            if (p == c->rank || pbase) c->opened.emplace_back(key, pbase);       // (the own handle: remembered as seen, nothing mapped)
        }
        if (p != c->rank && pbase) srcs[(size_t)p] = (const char *)pbase + m.offset;
    }
    if (any_new) {
        int32_t mine_ok = opened_ok ? 0 : 1;
        std::vector<int32_t> oks((size_t)c->world);
        rc = c->peer_cb(c->peer_user, &mine_ok, oks.data(), sizeof(int32_t));
        BDF_REQUIRE(rc == 0, BDF_ERR_HIP, "bdf_comm (peer copies): the host exchange function returned %d", rc);
        for (int p = 0; p < c->world; p++)
            if (oks[(size_t)p] != 0) {
                if (p != c->rank) bdf_set_error("bdf_comm (peer copies): rank %d could not open a peer's allocation", p);
                return BDF_ERR_HIP;
            }
    }
    int k = 0;
    for (int p = 0; p < c->world; p++) {
        if (p == c->rank) continue;
        hipStream_t st = c->pstreams[(size_t)k];
        BDF_HIP(hipStreamWaitEvent(st, c->peer_ready[(size_t)p * bdf_comm::RING + slot], 0));       // the owner's block is complete
        BDF_HIP(hipMemcpyAsync(region + (size_t)p * bytes, srcs[(size_t)p] + (size_t)p * bytes, bytes, hipMemcpyDeviceToDevice, st));
        BDF_HIP(hipEventRecord(c->pdone[(size_t)k], st));
        BDF_HIP(hipStreamWaitEvent(c->stream, c->pdone[(size_t)k], 0));                             // bdf_allgather_join waits for c->stream
        k++;
    }
    c->join_pending = true;
    c->seq++;
    c->peer_exchanges++;
    c->peer_bytes += (int64_t)bytes * (c->world - 1);
    return BDF_OK;
}
}  // namespace

extern "C" int bdf_comm_unique_id(void *id_out)
{
    BDF_REQUIRE(id_out, BDF_ERR_ARG, "bdf_comm_unique_id: NULL argument");
    int rc = load_rccl();
    if (rc) return rc;
    NcclId id;
    BDF_NCCL(g_rccl.get_id(&id));
    memcpy(id_out, &id, sizeof(id));
    return BDF_OK;
}

static int comm_new(bdf_ctx *ctx, int rank, int world, bdf_comm **out)
{
    BDF_REQUIRE(ctx && out && world >= 1 && rank >= 0 && rank < world, BDF_ERR_ARG, "bdf_comm_create: bad argument");
    BDF_HIP(hipSetDevice(ctx->device));
    bdf_comm *c = new bdf_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->nccl = nullptr; c->cb = nullptr; c->cb_user = nullptr;
    c->stream = nullptr; c->ev_rows = c->ev_done = nullptr;
    struct Guard { bdf_comm *c; ~Guard() { if (c) bdf_comm_destroy(c); } } guard{c};      // a failure below releases what exists so far
    BDF_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    BDF_HIP(hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming));
    BDF_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    guard.c = nullptr;
    *out = c;
    return BDF_OK;
}

extern "C" int bdf_comm_create(bdf_ctx *ctx, int rank, int world, const void *unique_id, bdf_comm **out)
{
    BDF_REQUIRE(unique_id, BDF_ERR_ARG, "bdf_comm_create: unique_id is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    bdf_comm *c;
    if ((rc = comm_new(ctx, rank, world, &c))) return rc;
    NcclId id;
    memcpy(&id, unique_id, sizeof(id));
    int e = g_rccl.init_rank(&c->nccl, world, id, rank);
    if (e != 0) {
        bdf_set_error("ncclCommInitRank failed: %s", g_rccl.err ? g_rccl.err(e) : "?");
        bdf_comm_destroy(c);
        return BDF_ERR_HIP;
    }
    *out = c;
    return BDF_OK;
}

extern "C" int bdf_comm_create_host(bdf_ctx *ctx, int rank, int world, bdf_exchange_fn fn, void *user, bdf_comm **out)
{
    BDF_REQUIRE(fn, BDF_ERR_ARG, "bdf_comm_create_host: fn is NULL");
    bdf_comm *c;
    int rc = comm_new(ctx, rank, world, &c);
    if (rc) return rc;
    c->cb = fn; c->cb_user = user;
    *out = c;
    return BDF_OK;
}

// Large exchanges by direct all-pairs copies (the header of this file).  fn: the caller's HOST all-gather (the control
// messages); min_bytes: exchanges of at least this much per rank take it.  Local: no collective here (the first exchange opens
// the ranks' interprocess events: that one is).
extern "C" int bdf_comm_enable_peer(bdf_comm *c, bdf_exchange_fn fn, void *user, size_t min_bytes)
{
    BDF_REQUIRE(c && fn, BDF_ERR_ARG, "bdf_comm_enable_peer: NULL argument");
    BDF_HIP(hipSetDevice(c->ctx->device));
    while ((int)c->pstreams.size() < c->world - 1) {
        hipStream_t st;
        hipEvent_t ev;
        BDF_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        c->pstreams.push_back(st);
        BDF_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->pdone.push_back(ev);
    }
    if (!c->nonce) {
        std::random_device rd;
        c->nonce = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 20) ^ (uint64_t)(uintptr_t)c;
        if (!c->nonce) c->nonce = 1;
    }
    c->peer_cb = fn; c->peer_user = user; c->peer_min_bytes = min_bytes;
    return BDF_OK;
}

extern "C" int bdf_comm_disable_peer(bdf_comm *c)
{
    BDF_REQUIRE(c, BDF_ERR_ARG, "bdf_comm_disable_peer: NULL argument");
    c->peer_cb = nullptr;
    return BDF_OK;
}

// (collective) one exchange by peer copies whatever its size, completed on return: the self-test a host runs before it lets the
// rows take this path -- buf holds world blocks of `bytes`, this rank's filled in
extern "C" int bdf_comm_peer_selftest(bdf_ctx *ctx, bdf_comm *c, void *buf, size_t bytes)
{
    BDF_REQUIRE(ctx && c && buf && bytes > 0, BDF_ERR_ARG, "bdf_comm_peer_selftest: bad argument");
    BDF_REQUIRE(c->peer_cb && c->world > 1, BDF_ERR_ARG, "bdf_comm_peer_selftest: peer copies are not enabled on this communicator");
    int rc = peer_allgather(ctx, c, (char *)buf, bytes);
    if (rc) return rc;
    if ((rc = bdf_allgather_join(ctx, c))) return rc;
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    // every rank's copies are complete before any rank reuses the buffer (the caller's, not a rotating one)
    int32_t mine = 0;
    std::vector<int32_t> all((size_t)c->world);
    rc = c->peer_cb(c->peer_user, &mine, all.data(), sizeof(int32_t));
    BDF_REQUIRE(rc == 0, BDF_ERR_HIP, "bdf_comm_peer_selftest: the host exchange function returned %d", rc);
    return BDF_OK;
}

// exchanges made by direct copies so far, and the bytes this rank pulled
extern "C" int bdf_comm_peer_stats(const bdf_comm *c, int64_t *exchanges, int64_t *bytes)
{
    BDF_REQUIRE(c && exchanges && bytes, BDF_ERR_ARG, "bdf_comm_peer_stats: NULL argument");
    *exchanges = c->peer_exchanges; *bytes = c->peer_bytes;
    return BDF_OK;
}

extern "C" int bdf_comm_destroy(bdf_comm *c)
{
    if (!c) return BDF_OK;
    (void)hipSetDevice(c->ctx->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl && g_rccl.destroy) g_rccl.destroy(c->nccl);
    for (auto &o : c->opened) if (o.second) (void)hipIpcCloseMemHandle(o.second);
    for (hipStream_t st : c->pstreams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t ev : c->pdone) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->peer_ready) if (ev) (void)hipEventDestroy(ev);
    for (int k = 0; k < bdf_comm::RING; k++) if (c->ready[k]) (void)hipEventDestroy(c->ready[k]);
    if (c->ev_rows) (void)hipEventDestroy(c->ev_rows);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return BDF_OK;
}

extern "C" int bdf_comm_size(const bdf_comm *c, int *rank, int *world)
{
    BDF_REQUIRE(c && rank && world, BDF_ERR_ARG, "bdf_comm_size: NULL argument");
    *rank = c->rank; *world = c->world;
    return BDF_OK;
}

// chunk `chunk` of `chunks` of the N x D factor matrix `sample` (N = chunks * world * cmax rows): every rank contributes its
// block [(chunk * world + rank) * cmax, + cmax) and receives the others', in place.  Ordered after the work enqueued so far
// on ctx's stream; runs on the communicator's stream; bdf_allgather_join makes ctx's stream wait for every exchange so far.
extern "C" int bdf_allgather_rows(bdf_ctx *ctx, bdf_comm *c, int D, int64_t N, double *sample, int chunk, int chunks)
{
    BDF_REQUIRE(ctx && c && sample, BDF_ERR_ARG, "bdf_allgather_rows: NULL argument");
    BDF_REQUIRE(chunks >= 1 && chunk >= 0 && chunk < chunks && N % ((int64_t)chunks * c->world) == 0, BDF_ERR_ARG,
                "bdf_allgather_rows: %lld rows are not %d chunks x %d ranks x cmax", (long long)N, chunks, c->world);
    const int64_t cmax = N / ((int64_t)chunks * c->world);
    const size_t count = (size_t)cmax * (size_t)D;                         // doubles per rank
    double *region = sample + (size_t)chunk * (size_t)c->world * count;
    if (count == 0 || (c->world == 1 && !c->nccl)) return BDF_OK;
    if (c->peer_cb && c->world > 1 && count * sizeof(double) >= c->peer_min_bytes) return peer_allgather(ctx, c, (char *)region, count * sizeof(double));
    if (c->nccl) {       // (a one-rank RCCL communicator still goes through ncclAllGather: the call path is exercised on one GPU)
        BDF_HIP(hipEventRecord(c->ev_rows, ctx->stream));
        BDF_HIP(hipStreamWaitEvent(c->stream, c->ev_rows, 0));
        BDF_NCCL(g_rccl.all_gather(region + (size_t)c->rank * count, region, count, 8 /* ncclFloat64 */, c->nccl, c->stream));
        return BDF_OK;
    }
    // host transport (test rigs): synchronous
    const size_t bytes = count * sizeof(double);
    c->hsend.resize(bytes); c->hrecv.resize(bytes * (size_t)c->world);
    BDF_HIP(hipMemcpyAsync(c->hsend.data(), region + (size_t)c->rank * count, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    int rc = c->cb(c->cb_user, c->hsend.data(), c->hrecv.data(), bytes);
    BDF_REQUIRE(rc == 0, BDF_ERR_HIP, "bdf_allgather_rows: the host exchange function returned %d", rc);
    BDF_HIP(hipMemcpyAsync(region, c->hrecv.data(), bytes * (size_t)c->world, hipMemcpyHostToDevice, ctx->stream));
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    return BDF_OK;
}

// in-place all-gather of `bytes` bytes per rank: rank r's block sits at buf + r * bytes.  Ordered after the work enqueued so far on
// ctx's stream, runs on the communicator's stream (bdf_allgather_join: ctx's stream waits for it); host transport: synchronous.
extern "C" int bdf_allgather_block(bdf_ctx *ctx, bdf_comm *c, void *buf, size_t bytes)
{
    BDF_REQUIRE(ctx && c && buf, BDF_ERR_ARG, "bdf_allgather_block: NULL argument");
    if (bytes == 0 || (c->world == 1 && !c->nccl)) return BDF_OK;
    char *b = (char *)buf;
    // (never by peer copies: this call's users exchange through ONE buffer, call after call -- nothing like the rows' rotating
    // buffers keeps an owner from rewriting a block a slow peer is still reading)
    if (c->nccl) {
        BDF_HIP(hipEventRecord(c->ev_rows, ctx->stream));
        BDF_HIP(hipStreamWaitEvent(c->stream, c->ev_rows, 0));
        BDF_NCCL(g_rccl.all_gather(b + (size_t)c->rank * bytes, b, bytes, 1 /* ncclUint8 */, c->nccl, c->stream));
        return BDF_OK;
    }
    c->hsend.resize(bytes); c->hrecv.resize(bytes * (size_t)c->world);
    BDF_HIP(hipMemcpyAsync(c->hsend.data(), b + (size_t)c->rank * bytes, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    int rc = c->cb(c->cb_user, c->hsend.data(), c->hrecv.data(), bytes);
    BDF_REQUIRE(rc == 0, BDF_ERR_HIP, "bdf_allgather_block: the host exchange function returned %d", rc);
    BDF_HIP(hipMemcpyAsync(b, c->hrecv.data(), bytes * (size_t)c->world, hipMemcpyHostToDevice, ctx->stream));
    BDF_HIP(hipStreamSynchronize(ctx->stream));
    return BDF_OK;
}

extern "C" int bdf_allgather_join(bdf_ctx *ctx, bdf_comm *c)
{
    BDF_REQUIRE(ctx && c, BDF_ERR_ARG, "bdf_allgather_join: NULL argument");
    if (!c->nccl && !c->join_pending) return BDF_OK;
    c->join_pending = false;
    BDF_HIP(hipEventRecord(c->ev_done, c->stream));
    BDF_HIP(hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
    return BDF_OK;
}
