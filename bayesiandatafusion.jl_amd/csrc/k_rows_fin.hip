// k_rows_fin.hip -- the FINISH of K1 as a launch of its own, FOUR ROWS PER WAVE (16 < D <= 32).
//
// The wave-per-row kernel (k_sample_rows.hip) keeps a row's 32 x 32 system in the accumulator layout of the matrix
// instructions and factors it there: ~800 vector instructions per row, a dependent chain of 32 steps, and -- because the
// kernel's register count is the factorisation's -- seven waves per SIMD for the gathers too.  Measured (DESIGN.md
// section 4): the finish is more of the launch's FP64-pipe time than the accumulation (16 against 11 us at MovieLens's size).
// Here a launch is TWO kernels: k_rows<..., SYS = true> accumulates every row (and sums the pieces of split rows) and leaves
// the row's system -- alpha S and alpha W r in the partial-slot format -- in a slab; this kernel then takes the rows four to
// a wave.  Every 16-lane row of the wave owns one entity row; lane j of it holds COLUMNS j and 16 + j of the index-reversed
// system P~ = Lambda~ + alpha S~ (all 32 rows of each: the unfinished part stays symmetric, so the multiplier of column c
// in step k is the lane's own entry (k, c) -- no transposition, no LDS) and entries j, 16 + j of b~.  The LDL' factorisation
// with the forward solve riding along as row 32, and the backward solve, are v_fmac_f64_dpp row_newbcast instructions: lane
// k % 16 of each lane row is the broadcast source of step k.  ~1,250 vector instructions per FOUR rows, every one of them
// doing the work of four rows.
// Same arithmetic contract as k_rows / k_rows_small: the sample is x~ = L~^-T (D^-1 L~^-1 b~ + D^-1/2 z~) of P~ = L~ D L~',
// column c drawing number D - 1 - c of the row's stream (oracle: orc_sample_rows).  The factorisation is unique, so the
// sample equals the wave-per-row kernel's to rounding (not to the last bit: the sums run in another order).
#include "bdf_common.h"
#include "dpp_rows32.h"

namespace {

typedef bdf_fin_item FinItem;

// the row's system from the slab plus the prior's image, eight rows at a time (every load in flight at once would take the
// registers of a second system)
template <int S, int I, int I1, bool POLLED>
__device__ __forceinline__ void fin_load8(double (&A)[33], const double *sys, const double *prior, int j)
{
    if constexpr (I < I1) {
        const int o = sys_off<S, I>(j);
        if constexpr (POLLED) A[I] = sys[o] + __hip_atomic_load(prior + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else A[I] = sys[o] + prior[o];
        fin_load8<S, I + 1, I1, POLLED>(A, sys, prior, j);
    }
}
template <int S, int I, int DR, bool POLLED>
__device__ __forceinline__ void fin_load(double (&A)[33], const double *sys, const double *prior, int j)
{
    if constexpr (I < DR) {
        fin_load8<S, I, (I + 8 < DR ? I + 8 : DR), POLLED>(A, sys, prior, j);
        asm volatile("" ::: "memory");
        fin_load<S, I + 8, DR, POLLED>(A, sys, prior, j);
    }
}

// one more slot of the row (a piece of a long row, another relation's part) added to the system, eight rows at a time
template <int S, int I, int I1>
__device__ __forceinline__ void fin_add16(double (&A)[33], const double *sys, int j)
{
    if constexpr (I < I1) {
        A[I] += sys[sys_off<S, I>(j)];
        fin_add16<S, I + 1, I1>(A, sys, j);
    }
}
template <int S, int I, int DR>
__device__ __forceinline__ void fin_add(double (&A)[33], const double *sys, int j)
{
    if constexpr (I < DR) {
        fin_add16<S, I, (I + 8 < DR ? I + 8 : DR)>(A, sys, j);
        asm volatile("" ::: "memory");
        fin_add<S, I + 8, DR>(A, sys, j);
    }
}

constexpr int FIN_PSZ = 3 * 4 * 64 + 2 * 16;          // Geo<32>::PSZ

template <int DR, bool POLLED>
__global__ __launch_bounds__(64, 2) void k_rows_fin(SampleArgs a, const FinItem *items, int64_t n_items, const double *slab)
{
    const int lane = threadIdx.x & 63, j = lane & 15;
    const int64_t w = blockIdx.x;
    if (w * 4 >= n_items) return;
    const FinItem it = items[w * 4 + (lane >> 4)];
    const bool live = it.row >= 0;
    const int D = a.D;
    const int ec0 = D - 1 - j, ec1 = D - 17 - j;          // natural index of the reversed elements j, 16 + j (negative: padding)
    const int n0 = ec0 >= 0 ? ec0 : 0, n1 = ec1 >= 0 ? ec1 : 0;
    // normals: lane p of the lane row draws pair p of the row's stream (numbers 2p, 2p + 1); column c wants number D - 1 - c
    double ze = 0.0, zo = 0.0;
    if (live && 2 * j < D) bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)it.orig, (uint32_t)j, ze, zo);
    const int base = lane & 48;
    const double ze0 = __shfl(ze, base + (n0 >> 1)), zo0 = __shfl(zo, base + (n0 >> 1));
    const double ze1 = __shfl(ze, base + (n1 >> 1)), zo1 = __shfl(zo, base + (n1 >> 1));
    const double z0 = ec0 >= 0 ? ((n0 & 1) ? zo0 : ze0) : 0.0, z1 = ec1 >= 0 ? ((n1 & 1) ? zo1 : ze1) : 0.0;
    asm volatile("" ::: "memory");

    // the row's system (after the normals: their arithmetic needs ~60 registers of its own) plus the prior
    const int n_slots = live ? it._pad : 0;              // (a row without observations has none: `sys` is then the slab's zero slot)
    const double *sys = slab + (int64_t)(live ? it.sys : 0) * FIN_PSZ;
    double A0[33], A1[33];
    const int ob0 = 3 * 4 * 64 + j, ob1 = ob0 + 16;
    const int64_t pb = a.mu_is_matrix && live ? (int64_t)it.row * D : 0;
    if constexpr (POLLED) {
        // launched without waiting for the hyperprior draw (bdf_gibbs_sweep): poll its flag here and read the pack past the
        // non-coherent caches, as k_rows does
        int spins = 0;
        while ((int32_t)(__hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ready_want) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 22)) { if (lane == 0) atomicOr_system(a.flag, 16); break; }
        }
        A0[32] = sys[ob0] + (ec0 >= 0 ? __hip_atomic_load(a.prior_b + pb + n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0);
        A1[32] = sys[ob1] + (ec1 >= 0 ? __hip_atomic_load(a.prior_b + pb + n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0);
    } else {
        A0[32] = sys[ob0] + (ec0 >= 0 ? a.prior_b[pb + n0] : 0.0);
        A1[32] = sys[ob1] + (ec1 >= 0 ? a.prior_b[pb + n1] : 0.0);
    }
    fin_load<0, 0, DR, POLLED>(A0, sys, a.prior_c, j);
    fin_load<1, 0, DR, POLLED>(A1, sys, a.prior_c, j);
    // the other slots of rows that have several, in slot order (the rows of a wave were sorted to have like counts)
    int nmax = n_slots;
    nmax = max(nmax, __shfl_xor(nmax, 16));
    nmax = max(nmax, __shfl_xor(nmax, 32));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    for (int sl = 1; sl < nmax; sl++) {
        if (sl < n_slots) {
            const double *sp = sys + (int64_t)sl * FIN_PSZ;
            A0[32] += sp[ob0];
            A1[32] += sp[ob1];
            fin_add<0, 0, DR>(A0, sp, j);
            fin_add<1, 0, DR>(A1, sp, j);
        }
    }
    // (rows and columns D .. 31 of the system are the identity by construction: masked gathers, the prior's image)
    double d0 = 1.0, d1 = 1.0;
    fin_factor<DR, 0>(A0, A1, d0, d1, j);
    if (live && ((ec0 >= 0 && !(d0 > 0.0)) || (ec1 >= 0 && !(d1 > 0.0)))) atomicOr_system(a.flag, 1);      // not positive definite
    const double rd0 = fast_rcp(d0), rd1 = fast_rcp(d1);
    double y0 = fma(z0, fast_rsqrt(d0), A0[32] * rd0), y1 = fma(z1, fast_rsqrt(d1), A1[32] * rd1);
    fin_backward<DR - 1>(A0, A1, y0, y1, rd0, rd1, j);
    if (live && ec0 >= 0) a.out[(int64_t)it.row * D + ec0] = y0;
    if (live && ec1 >= 0) a.out[(int64_t)it.row * D + ec1] = y1;
}

}  // namespace

int bdf_fin_launch(bdf_ctx *ctx, const SampleArgs &a, const bdf_fin_item *fi, int64_t n_items, const double *slab, hipEvent_t e0, hipEvent_t e1)
{
    if (n_items <= 0) return BDF_OK;
    const dim3 grid((unsigned)((n_items + 3) / 4)), block(64);
    const int DR = (a.D + 3) / 4 * 4;
#define FIN_LAUNCH(DRV)                                                                                                                   \
    do {                                                                                                                                  \
        if (a.ready) hipExtLaunchKernelGGL((k_rows_fin<DRV, true>), grid, block, 0, ctx->stream, e0, e1, 0, a, fi, n_items, slab);        \
        else hipExtLaunchKernelGGL((k_rows_fin<DRV, false>), grid, block, 0, ctx->stream, e0, e1, 0, a, fi, n_items, slab);               \
    } while (0)
    if (DR <= 20) FIN_LAUNCH(20); else if (DR <= 24) FIN_LAUNCH(24); else if (DR <= 28) FIN_LAUNCH(28); else FIN_LAUNCH(32);
#undef FIN_LAUNCH
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}
