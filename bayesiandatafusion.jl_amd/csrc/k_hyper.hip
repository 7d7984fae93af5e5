// k_hyper.hip -- K6: hyperprior of an entity's latent rows.
//
//  bdf_hyper_sums   : N, sum_i U_i and U U' (src/sampling.jl:117-119) with U = sample - uhat (macau.jl:123),
//                     two-stage deterministic reduction.
//  bdf_hyper_sample : ConditionalNormalWishart (src/sampling.jl:116-127) + rand(::NormalWishart)
//                     (src/normal_wishart.jl:38-42) in ONE workgroup, so the D x D work never leaves the device.
//
// The Normal-Wishart draw, in the reference's terms:
//     W    = Tinv + UU' + b0 mu0 mu0' - beta_N mu_N mu_N'         (= inv(T_N))
//     Lam  = (L_T A)(L_T A)',  L_T = chol(T_N)' lower,  A = Bartlett matrix (A_aa = sqrt(chi2(nu_N - a)), A_ac ~ N(0,1), c < a)
//     mu   = mu_N + chol(inv(Lam) / beta_N)' z
// As in K1 both "Cholesky factor of an inverse" steps are obtained without forming the inverse: with W = U U'
// (U upper), L_T == U^-T; with Lam = U2 U2', chol(inv(Lam))' == U2^-T.  In index-reversed coordinates (~) these are
// ordinary lower Cholesky factors:  W~ = L~ L~',  Z~ = L~^-T (J A),  Lam~ = Z~ Z~',  Lam~ = L2~ L2~',
// mu~ = mu_N~ + L2~^-T z~ / sqrt(beta_N).
#include "bdf_common.h"
#define BDF_CHOL_LOOKAHEAD     // one lone workgroup: the next pivot's reciprocal ahead of the step (K1 is VALU-bound: off there)
#include "hyper_job.h"
#include <algorithm>

namespace {

template <int DP>
__global__ __launch_bounds__(HS_THREADS) void k_hyper_partial(int D, int64_t N, int64_t rows_per_block,
                                                               const double *__restrict__ sample,
                                                               const double *__restrict__ uhat, double *__restrict__ partial)
{
    __shared__ double red[3 * HGeo<DP>::PSZ];
    // small and on the sweep's critical path, usually beside a chip-filling K1 launch: ahead of it at the issue port -- but not the sums
    // of a large entity (thousands of workgroups beside the other entity's rows: they would only take the rows' issue slots)
    if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(3);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
    hyper_partial<DP, 4>(D, N, sample, uhat, r0, r1, partial + (int64_t)blockIdx.x * HGeo<DP>::PSZ, red, threadIdx.x);
}

// ---- stage 2: fixed-order sum of the partials -----------------------------------------------------------------
// 16 lanes per partial element: lane q sums blocks q, q+16, ... and the 16 sums are combined by a butterfly -- a fixed
// order (hyper_sum_element reproduces it on one thread), so the result does not depend on scheduling
template <int DP>
__global__ __launch_bounds__(256) void k_hyper_final(int D, int nblocks, const double *__restrict__ partial,
                                                     double *__restrict__ sumU, double *__restrict__ UUt)
{
    constexpr int PSZ = HGeo<DP>::PSZ;
    __builtin_amdgcn_s_setprio(3);
    const int q = threadIdx.x & 15;
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 4;
    double s = 0.0;
    if (e < PSZ)
        for (int b = q; b < nblocks; b += 16) s += partial[(int64_t)b * PSZ + e];
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (q != 0 || e >= PSZ) return;
    hyper_scatter<DP>(D, e, s, sumU, UUt);
}

// The random part of the draw does not depend on the data: the Bartlett matrix A (A_aa = sqrt(chi2(nu_N - a)),
// A_ac ~ N(0,1) for c < a, row-major D x D) and the D normals of the mean can be drawn while the rows are still being
// sampled (bdf_hyper_draws, one lane per entry over many workgroups), which takes the slowest part of k_hyper_sample --
// the gamma rejection loops -- off the sweep's critical path.  Same streams and values as the in-kernel draw.
__global__ __launch_bounds__(64) void k_hyper_draws(int D, double nu_N, uint64_t seed, uint32_t sweep, uint32_t entity_tag,
                                                    double *out)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e < D * D) {
        const int arow = e / D, c = e % D;
        double v = 0.0;
        if (c < arow) v = bdf_normal(seed, sweep, BDF_P_NW_NORMAL, entity_tag, (uint64_t)arow, c);
        else if (c == arow) v = sqrt(2.0 * bdf_gamma(seed, sweep, entity_tag, (uint64_t)arow, 0.5 * (nu_N - (double)arow)));
        out[e] = v;
    } else if (e < D * D + D) {
        out[e] = bdf_normal(seed, sweep, BDF_P_NW_MEAN, entity_tag, 0, e - D * D);
    }
}

// the same for several entities in one launch (blockIdx.y = entity): bdf_gibbs_sweep makes every entity's draws at the head of
// the iteration
struct DrawsBatch {
    int n;
    double nu_N[BDF_DRAWS_BATCH];
    uint32_t tag[BDF_DRAWS_BATCH];
    double *out[BDF_DRAWS_BATCH];
};
__global__ __launch_bounds__(64) void k_hyper_draws_batch(int D, DrawsBatch b, uint64_t seed, uint32_t sweep)
{
    const int j = blockIdx.y;
    const int e = blockIdx.x * 64 + threadIdx.x;
    const uint32_t entity_tag = b.tag[j];
    double *out = b.out[j];
    if (e < D * D) {
        const int arow = e / D, c = e % D;
        double v = 0.0;
        if (c < arow) v = bdf_normal(seed, sweep, BDF_P_NW_NORMAL, entity_tag, (uint64_t)arow, c);
        else if (c == arow) v = sqrt(2.0 * bdf_gamma(seed, sweep, entity_tag, (uint64_t)arow, 0.5 * (b.nu_N[j] - (double)arow)));
        out[e] = v;
    } else if (e < D * D + D) {
        out[e] = bdf_normal(seed, sweep, BDF_P_NW_MEAN, entity_tag, 0, e - D * D);
    }
}

// One workgroup of 256 threads (hyper_job.h: nw_draw).
template <int DP>
__global__ __launch_bounds__(256) void k_hyper_sample(NWArgs a)
{
    __shared__ __attribute__((aligned(16))) double lds[HGeo<DP>::NW_LDS];
    __builtin_amdgcn_s_setprio(3);      // one workgroup beside a chip-filling K1 launch: take the issue slots when ready
    nw_draw<DP>(a, lds, threadIdx.x, 256);
    if (a.ready) {                      // the pack (write-through stores) is complete: publish
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.ready, a.ready_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The whole chain of a small entity in ONE launch: workgroups 0 .. nblocks-1 are k_hyper_partial's, the last one is
// k_hyper_sample's -- dispatched after them (workgroups are dispatched in order), it waits for their count, adds the partials
// and draws.  One launch and one kernel boundary fewer per entity and iteration than sums -> draw.
struct ChainArgs {
    int D; int64_t N, rows_per_block; const double *sample, *uhat; double *partial; unsigned *count; int nblocks;
    // ndraw > 0: the first ndraw workgroups make the data-independent random part (k_hyper_draws' entries, 256 per workgroup)
    // into draws_out -- beside the partial sums instead of in a launch of their own on the same stream
    int ndraw; double nu_N; uint64_t seed; uint32_t sweep, tag; double *draws_out;
    // nullable: the launch was NOT ordered behind the rows' launch; the sums' workgroups wait until the 64 counters the row waves add to
    // (SampleArgs::done) sum to rows_target (bdf_gibbs_sweep: no event, no stream wait -- a wait for another stream's event costs the
    // waiting stream ~10 us even when the event completed long before)
    const uint32_t *rows_done; uint32_t rows_target; int *flag;
};
template <int DP>
__global__ __launch_bounds__(256) void k_hyper_chain(ChainArgs c, NWArgs a)
{
    constexpr int LDS_D = (3 * HGeo<DP>::PSZ > HGeo<DP>::NW_LDS) ? 3 * HGeo<DP>::PSZ : HGeo<DP>::NW_LDS;
    __shared__ __attribute__((aligned(16))) double lds[LDS_D];
    __builtin_amdgcn_s_setprio(3);
    if ((int)blockIdx.x < c.ndraw) {
        const int D = c.D, e = blockIdx.x * 256 + threadIdx.x;
        if (e < D * D + D) {
            double v = 0.0;
            if (e < D * D) {
                const int arow = e / D, col = e % D;
                if (col < arow) v = bdf_normal(c.seed, c.sweep, BDF_P_NW_NORMAL, c.tag, (uint64_t)arow, col);
                else if (col == arow) v = sqrt(2.0 * bdf_gamma(c.seed, c.sweep, c.tag, (uint64_t)arow, 0.5 * (c.nu_N - (double)arow)));
            } else v = bdf_normal(c.seed, c.sweep, BDF_P_NW_MEAN, c.tag, 0, e - D * D);
            __hip_atomic_store(c.draws_out + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // write-through
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(c.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int pb = (int)blockIdx.x - c.ndraw;
    if (pb < c.nblocks) {
        const int64_t r0 = (int64_t)pb * c.rows_per_block;
        const int64_t r1 = r0 + c.rows_per_block < c.N ? r0 + c.rows_per_block : c.N;
        if (c.rows_done) {
            if (threadIdx.x < 64) {
                int spins = 0;
                for (;;) {
                    uint32_t v = __hip_atomic_load(c.rows_done + BDF_DONE_STRIDE * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
                    if ((int32_t)(v - c.rows_target) >= 0) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 22)) { if (threadIdx.x == 0) atomicOr_system(c.flag, 16); break; }      // bounded: a bug must not hang the device
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
#ifdef BDF_HYPER_STAMPS
        if (threadIdx.x == 0 && (pb == 0 || pb == c.nblocks - 1)) g_hstamps[pb == 0 ? 10 : 12] = __builtin_amdgcn_s_memrealtime();
#endif
        hyper_partial<DP, 4>(c.D, c.N, c.sample, c.uhat, r0, r1, c.partial + (int64_t)pb * HGeo<DP>::PSZ, lds, threadIdx.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's write-through stores of the partial have completed
        __syncthreads();
#ifdef BDF_HYPER_STAMPS
        if (threadIdx.x == 0 && (pb == 0 || pb == c.nblocks - 1)) g_hstamps[pb == 0 ? 11 : 13] = __builtin_amdgcn_s_memrealtime();
#endif
        if (threadIdx.x == 0) __hip_atomic_fetch_add(c.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#ifdef BDF_HYPER_STAMPS
    if (threadIdx.x == 0) g_hstamps[8] = __builtin_amdgcn_s_memrealtime();
#endif
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(c.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(c.nblocks + c.ndraw)) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 24)) { atomicOr_system(a.flag, 16); break; }       // bounded: a bug must not hang the device
        }
        __hip_atomic_store(c.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#ifdef BDF_HYPER_STAMPS
    if (threadIdx.x == 0) g_hstamps[9] = __builtin_amdgcn_s_memrealtime();
#endif
    nw_draw<DP>(a, lds, threadIdx.x, 256);
    if (a.ready) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.ready, a.ready_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

extern "C" int bdf_hyper_sums(bdf_ctx *ctx, int D, int64_t N, const double *sample, const double *uhat,
                              double *sumU, double *UUt)
{
    const bool fuse_asked = ctx && ctx->hyper_fuse;       // one call only, whatever happens below
    if (ctx) { ctx->hyper_fuse = false; ctx->hyper_partial = nullptr; ctx->hyper_chain = false; }
    BDF_REQUIRE(ctx && sample && sumU && UUt, BDF_ERR_ARG, "bdf_hyper_sums: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_sums: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    BDF_REQUIRE(N >= 0, BDF_ERR_ARG, "bdf_hyper_sums: N < 0");
    // HS_ROWS rows per workgroup, at most 2048 workgroups (a very large entity gives each several chunks)
    const int64_t chunks = std::max<int64_t>(1, (N + HS_ROWS - 1) / HS_ROWS);
    // the caller enqueues the draw next and lets it add the partials (bdf_gibbs_sweep): at most 16 of them, for entities small
    // enough that 16 workgroups read them quickly
    const bool fuse = fuse_asked && N <= 16384;
    static const bool one_launch_ = !(getenv("BDF_HYPER_CHAIN") && atoi(getenv("BDF_HYPER_CHAIN")) == 0);
    // the one-launch chain (k_hyper_chain): its workgroups -- the draws' (D^2 + D entries, 256 each), the partial sums', the last one --
    // should all be resident at once on the stream's CUs (two workgroups of 255 registers per CU), or the partial sums take two
    // rounds and the last workgroup gets its slot when the first round ends (9.3 us of the chain at D = 32).  The count comes from a
    // NOMINAL 16 slots (8 reserved CUs), not from the context's: the number of partials decides the order of the sums, and the
    // sampled values must not depend on BDF_RESERVE_CUS or the CU count (k_sample_rows.hip's rule for the rows' cuts)
    const int wg_slots = 16;
    const int max_part = (fuse && one_launch_) ? std::max(8, std::min(16, wg_slots - 1 - (D * D + D + 255) / 256)) : 16;
    const int64_t rpb = HS_ROWS * ((chunks + (fuse ? max_part - 1 : 2047)) / (fuse ? max_part : 2048));
    const int nblocks = (int)std::max<int64_t>(1, (N + rpb - 1) / rpb);
    const int DP = D <= 16 ? 16 : (D <= 32 ? 32 : 64);
    const int psz = DP == 16 ? HGeo<16>::PSZ : (DP == 32 ? HGeo<32>::PSZ : HGeo<64>::PSZ);
    void *scratch;
    int rc = bdf_scratch(ctx, (size_t)nblocks * psz * sizeof(double), &scratch);
    if (rc) return rc;
    double *part = (double *)scratch;
    const bool one_launch = one_launch_;
    if (fuse && one_launch) {
        // left to the bdf_hyper_sample that follows: one launch for the whole chain (k_hyper_chain)
        ctx->hyper_chain = true;
        ctx->hyper_chain_D = D; ctx->hyper_chain_N = N; ctx->hyper_chain_rpb = rpb; ctx->hyper_chain_sample = sample; ctx->hyper_chain_uhat = uhat;
        ctx->hyper_partial = part; ctx->hyper_nblocks = nblocks; ctx->hyper_sumU = sumU; ctx->hyper_UUt = UUt;
        return BDF_OK;
    }
    const dim3 fgrid((psz + 15) / 16);
    if (DP == 16) {
        hipExtLaunchKernelGGL(k_hyper_partial<16>, dim3(nblocks), dim3(HS_THREADS), 0, ctx->stream, ctx->time_h_start, nullptr, 0, D, N, rpb, sample, uhat, part);
        if (!fuse) hipLaunchKernelGGL(k_hyper_final<16>, fgrid, dim3(256), 0, ctx->stream, D, nblocks, (const double *)part, sumU, UUt);
    } else if (DP == 32) {
        hipExtLaunchKernelGGL(k_hyper_partial<32>, dim3(nblocks), dim3(HS_THREADS), 0, ctx->stream, ctx->time_h_start, nullptr, 0, D, N, rpb, sample, uhat, part);
        if (!fuse) hipLaunchKernelGGL(k_hyper_final<32>, fgrid, dim3(256), 0, ctx->stream, D, nblocks, (const double *)part, sumU, UUt);
    } else {
        hipExtLaunchKernelGGL(k_hyper_partial<64>, dim3(nblocks), dim3(HS_THREADS), 0, ctx->stream, ctx->time_h_start, nullptr, 0, D, N, rpb, sample, uhat, part);
        if (!fuse) hipLaunchKernelGGL(k_hyper_final<64>, fgrid, dim3(256), 0, ctx->stream, D, nblocks, (const double *)part, sumU, UUt);
    }
    ctx->time_h_start = nullptr;
    ctx->hyper_partial = fuse ? part : nullptr;
    ctx->hyper_nblocks = fuse ? nblocks : 0;
    ctx->hyper_sumU = sumU; ctx->hyper_UUt = UUt;
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

// Several ranks (the reference sums on the master, src/sampling.jl:117-119): every rank adds the rows IT OWNS -- with
// the layout of bdf_layout_build chunk c of rank p is the contiguous block of positions [(c P + p) cmax, + cmax) -- and the
// ranks' D + D^2 partial sums are gathered and added in rank order (bdf_sum_ranks): the same bits on every rank, and the
// reduction's work shrinks with the number of ranks instead of being repeated over the whole replica on each of them.
extern "C" int bdf_hyper_sums_ranks(bdf_ctx *ctx, bdf_comm *comm, int D, int64_t N, int chunks, const double *sample, const double *uhat,
                                    double *sumU, double *UUt)
{
    int rank = 0, world = 1, rc;
    if (comm && (rc = bdf_comm_size(comm, &rank, &world))) return rc;
    // (BDF_FORCE_COMM: the ranks' path with ONE rank too -- the soak of the schedule with RCCL's kernels on the device)
    static const bool force = getenv("BDF_FORCE_COMM") != nullptr;
    if (world <= 1 && !(force && comm)) return bdf_hyper_sums(ctx, D, N, sample, uhat, sumU, UUt);
    if (ctx) { ctx->hyper_fuse = false; ctx->hyper_partial = nullptr; ctx->hyper_chain = false; }
    BDF_REQUIRE(ctx && sample && sumU && UUt, BDF_ERR_ARG, "bdf_hyper_sums_ranks: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_sums_ranks: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    BDF_REQUIRE(chunks >= 1 && N >= 0 && N % ((int64_t)chunks * world) == 0, BDF_ERR_ARG,
                "bdf_hyper_sums_ranks: %lld rows are not %d chunks x %d ranks x cmax", (long long)N, chunks, world);
    const int64_t cmax = N / ((int64_t)chunks * world);
    const int64_t nch = std::max<int64_t>(1, (cmax + HS_ROWS - 1) / HS_ROWS);
    const int64_t per_chunk_cap = std::max<int64_t>(1, 2048 / chunks);
    const int64_t rpb = HS_ROWS * ((nch + per_chunk_cap - 1) / per_chunk_cap);
    const int nb = (int)std::max<int64_t>(1, (cmax + rpb - 1) / rpb);              // workgroups per chunk
    const int DP = D <= 16 ? 16 : (D <= 32 ? 32 : 64);
    const int psz = DP == 16 ? HGeo<16>::PSZ : (DP == 32 ? HGeo<32>::PSZ : HGeo<64>::PSZ);
    const size_t pack = (size_t)D + (size_t)D * D;
    void *scratch;
    if ((rc = bdf_scratch(ctx, ((size_t)nb * chunks * psz + pack * ((size_t)world + 1)) * sizeof(double), &scratch))) return rc;
    double *part = (double *)scratch, *mine = part + (size_t)nb * chunks * psz, *gathered = mine + pack;
    for (int c = 0; c < chunks; c++) {
        const int64_t r0 = ((int64_t)c * world + rank) * cmax;
        const double *sp = sample + r0 * D, *up = uhat ? uhat + r0 * D : nullptr;
        double *pp = part + (size_t)c * nb * psz;
        hipEvent_t e0 = c == 0 ? ctx->time_h_start : nullptr;
        if (DP == 16) hipExtLaunchKernelGGL(k_hyper_partial<16>, dim3(nb), dim3(HS_THREADS), 0, ctx->stream, e0, nullptr, 0, D, cmax, rpb, sp, up, pp);
        else if (DP == 32) hipExtLaunchKernelGGL(k_hyper_partial<32>, dim3(nb), dim3(HS_THREADS), 0, ctx->stream, e0, nullptr, 0, D, cmax, rpb, sp, up, pp);
        else hipExtLaunchKernelGGL(k_hyper_partial<64>, dim3(nb), dim3(HS_THREADS), 0, ctx->stream, e0, nullptr, 0, D, cmax, rpb, sp, up, pp);
    }
    ctx->time_h_start = nullptr;
    const dim3 fgrid((psz + 15) / 16);
    if (DP == 16) hipLaunchKernelGGL(k_hyper_final<16>, fgrid, dim3(256), 0, ctx->stream, D, nb * chunks, (const double *)part, mine, mine + D);
    else if (DP == 32) hipLaunchKernelGGL(k_hyper_final<32>, fgrid, dim3(256), 0, ctx->stream, D, nb * chunks, (const double *)part, mine, mine + D);
    else hipLaunchKernelGGL(k_hyper_final<64>, fgrid, dim3(256), 0, ctx->stream, D, nb * chunks, (const double *)part, mine, mine + D);
    BDF_HIP(hipGetLastError());
    if ((rc = bdf_sum_ranks_into(ctx, comm, mine, (int64_t)pack, gathered))) return rc;
    BDF_HIP(hipMemcpyAsync(sumU, mine, (size_t)D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    BDF_HIP(hipMemcpyAsync(UUt, mine + D, (size_t)D * D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ctx->hyper_sumU = sumU; ctx->hyper_UUt = UUt;
    return BDF_OK;
}

extern "C" int bdf_hyper_draws(bdf_ctx *ctx, int D, int64_t N, double nu, uint32_t entity_tag, double *draws_out)
{
    BDF_REQUIRE(ctx && draws_out, BDF_ERR_ARG, "bdf_hyper_draws: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_draws: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    const int total = D * D + D;
    hipLaunchKernelGGL(k_hyper_draws, dim3((total + 63) / 64), dim3(64), 0, ctx->stream, D, nu + (double)N, ctx->seed,
                       ctx->sweep_host, entity_tag, draws_out);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

// bdf_hyper_draws for n <= BDF_DRAWS_BATCH entities in one launch (internal: bdf_gibbs_sweep)
int bdf_hyper_draws_batch(bdf_ctx *ctx, int D, int n, const int64_t *N, const double *nu, const uint32_t *entity_tag, double *const *draws_out)
{
    BDF_REQUIRE(ctx && n >= 1 && n <= BDF_DRAWS_BATCH && N && nu && entity_tag && draws_out, BDF_ERR_ARG, "bdf_hyper_draws_batch: bad argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_draws_batch: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    DrawsBatch b;
    b.n = n;
    for (int j = 0; j < n; j++) { b.nu_N[j] = nu[j] + (double)N[j]; b.tag[j] = entity_tag[j]; b.out[j] = draws_out[j]; }
    const int total = D * D + D;
    hipLaunchKernelGGL(k_hyper_draws_batch, dim3((total + 63) / 64, n), dim3(64), 0, ctx->stream, D, b, ctx->seed, ctx->sweep_host);
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

extern "C" int bdf_prior_pack_doubles(int D)
{
    if (D < 1 || D > BDF_MAX_D) return 0;
    const int DB = (D <= 16 ? 16 : (D <= 32 ? 32 : 64)) / 16;
    return D + DB * (DB + 1) / 2 * 4 * 64;
}

extern "C" int bdf_hyper_sample(bdf_ctx *ctx, int D, int64_t N, const double *sumU, const double *UUt,
                                const double *mu0, double b0, const double *Tinv, double nu, uint32_t entity_tag,
                                double *mu_out, double *Lambda_out, double *params_out, double *prior_pack_out,
                                const double *draws)
{
    BDF_REQUIRE(ctx && sumU && UUt && mu0 && Tinv && mu_out && Lambda_out, BDF_ERR_ARG, "bdf_hyper_sample: NULL argument");
    BDF_REQUIRE(D >= 1 && D <= BDF_MAX_D, BDF_ERR_ARG, "bdf_hyper_sample: num_latent=%d must be in 1..%d", D, BDF_MAX_D);
    if (ctx->hyper_chain_draws && !(ctx->hyper_chain && ctx->hyper_partial)) {
        // the caller left the random part to the chain's launch, and there is no chain after all: make it now, same stream
        const int total = D * D + D;
        hipLaunchKernelGGL(k_hyper_draws, dim3((total + 63) / 64), dim3(64), 0, ctx->stream, D, nu + (double)N, ctx->seed,
                           ctx->sweep_host, entity_tag, ctx->hyper_chain_draws);
        BDF_HIP(hipGetLastError());
        ctx->hyper_chain_draws = nullptr;
    }
    if (draws == nullptr) {
        // no draws made ahead (bdf_hyper_draws): make them now, same stream, into the context's scratch
        void *sc;
        int rc = bdf_scratch(ctx, ((size_t)D * D + D) * sizeof(double), &sc);
        if (rc) return rc;
        const int total = D * D + D;
        hipLaunchKernelGGL(k_hyper_draws, dim3((total + 63) / 64), dim3(64), 0, ctx->stream, D, nu + (double)N, ctx->seed,
                           ctx->sweep_host, entity_tag, (double *)sc);
        BDF_HIP(hipGetLastError());
        draws = (const double *)sc;
    }
    NWArgs a;
    a.D = D; a.N = (double)N; a.sumU = sumU; a.UUt = UUt; a.mu0 = mu0; a.Tinv = Tinv; a.b0 = b0; a.nu = nu;
    a.mu_out = mu_out; a.Lambda_out = Lambda_out; a.params_out = params_out; a.pack_out = prior_pack_out; a.draws = draws;
    a.flag = ctx->flag_dev;
    a.ready = ctx->hyper_ready; a.ready_value = ctx->hyper_ready_value;
    ctx->hyper_ready = nullptr;
    a.partial = nullptr; a.nblocks = 0; a.sumU_w = a.UUt_w = nullptr;
    static const int mean_ref = getenv("BDF_HYPER_MEAN") && !strcmp(getenv("BDF_HYPER_MEAN"), "reference");
    a.mean_ref = mean_ref;
    if (ctx->hyper_partial) {
        // the partials of the bdf_hyper_sums call just before (same stream, fused mode): they live in the context's scratch
        BDF_REQUIRE(draws != nullptr && sumU == ctx->hyper_sumU && UUt == ctx->hyper_UUt, BDF_ERR_ARG,
                    "bdf_hyper_sample: fused sums need the draws made ahead and the sums' own output buffers");
        a.partial = ctx->hyper_partial; a.nblocks = ctx->hyper_nblocks; a.sumU_w = ctx->hyper_sumU; a.UUt_w = ctx->hyper_UUt;
        ctx->hyper_partial = nullptr;
    }
    if (ctx->hyper_chain && a.partial) {
        ctx->hyper_chain = false;
        if (!ctx->hyper_count) {
            BDF_HIP(hipMalloc((void **)&ctx->hyper_count, sizeof(unsigned)));
            BDF_HIP(hipMemsetAsync(ctx->hyper_count, 0, sizeof(unsigned), ctx->stream));
        }
        ChainArgs c;
        c.D = ctx->hyper_chain_D; c.N = ctx->hyper_chain_N; c.rows_per_block = ctx->hyper_chain_rpb; c.sample = ctx->hyper_chain_sample;
        c.uhat = ctx->hyper_chain_uhat; c.partial = const_cast<double *>(a.partial); c.count = ctx->hyper_count; c.nblocks = a.nblocks;
        c.ndraw = 0; c.nu_N = 0.0; c.seed = 0; c.sweep = 0; c.tag = 0; c.draws_out = nullptr;
        c.rows_done = ctx->hyper_wait; c.rows_target = ctx->hyper_wait_target; c.flag = ctx->flag_dev;
        ctx->hyper_wait = nullptr;
        if (ctx->hyper_chain_draws) {
            BDF_REQUIRE(ctx->hyper_chain_draws == draws, BDF_ERR_ARG, "bdf_hyper_sample: the chain's draws go to another buffer than the one the draw reads");
            c.ndraw = (D * D + D + 255) / 256; c.nu_N = nu + (double)N; c.seed = ctx->seed; c.sweep = ctx->sweep_host; c.tag = entity_tag;
            c.draws_out = ctx->hyper_chain_draws;
            ctx->hyper_chain_draws = nullptr;
        }
        const dim3 grid((unsigned)(a.nblocks + c.ndraw) + 1);
        if (D <= 16) hipExtLaunchKernelGGL(k_hyper_chain<16>, grid, dim3(256), 0, ctx->stream, ctx->time_h_start, ctx->time_h_stop, 0, c, a);
        else if (D <= 32) hipExtLaunchKernelGGL(k_hyper_chain<32>, grid, dim3(256), 0, ctx->stream, ctx->time_h_start, ctx->time_h_stop, 0, c, a);
        else hipExtLaunchKernelGGL(k_hyper_chain<64>, grid, dim3(256), 0, ctx->stream, ctx->time_h_start, ctx->time_h_stop, 0, c, a);
        ctx->time_h_start = ctx->time_h_stop = nullptr;
        BDF_HIP(hipGetLastError());
        return BDF_OK;
    }
    BDF_REQUIRE(!ctx->hyper_wait, BDF_ERR_ARG, "bdf_hyper_sample: a hand-over by counter needs the one-launch chain");
    if (D <= 16) hipExtLaunchKernelGGL(k_hyper_sample<16>, dim3(1), dim3(256), 0, ctx->stream, nullptr, ctx->time_h_stop, 0, a);
    else if (D <= 32) hipExtLaunchKernelGGL(k_hyper_sample<32>, dim3(1), dim3(256), 0, ctx->stream, nullptr, ctx->time_h_stop, 0, a);
    else hipExtLaunchKernelGGL(k_hyper_sample<64>, dim3(1), dim3(256), 0, ctx->stream, nullptr, ctx->time_h_stop, 0, a);
    ctx->time_h_stop = nullptr;
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}

#ifdef BDF_HYPER_STAMPS
extern "C" int bdf_debug_hyper_stamps(unsigned long long *host16)
{
    BDF_HIP(hipDeviceSynchronize());
    BDF_HIP(hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_hstamps), sizeof(unsigned long long) * 16));
    return BDF_OK;
}
#endif
