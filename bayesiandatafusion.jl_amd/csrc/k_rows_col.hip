// k_rows_col.hip -- K1c: the latent rows of ONE two-mode relation at 16 < D <= 32, FOUR ROWS PER WAVE from the first
// observation to the sample, no matrix instructions, no slab for rows of up to 4 T observations.
//
// Replaces sample_user_basic (src/sampling.jl:200-212) for every row of an entity at once (sample_latent_all2!
// :149-172) -- the same function of z as k_rows (k_sample_rows.hip): x~ = L~^-T (D^-1 L~^-1 b~ + D^-1/2 z~) of the
// index-reversed system P~ = Lambda~ + alpha S~ = L~ D L~', column c drawing number D - 1 - c of the row's stream.
//
// Why a second structure.  The wave-per-row kernel keeps a row's system in the accumulator layout of the matrix
// instructions: ~800 vector instructions per row for the factorisation and the solves at a fifth of the lanes' rate, one
// generation of ~6,700 waves whose SIMDs end when their most loaded one does, and a last wave per SIMD that factors alone
// at the rate a lone wave issues fp64 instructions (57 k pipe cycles in a 95 k-cycle launch).  Here
// every 16-lane row of a wave owns one entity row -- or one PIECE of a longer one -- and lane j of it holds COLUMNS j and
// 16 + j of the index-reversed system in full (dpp_rows32.h), FROM THE ACCUMULATION ON:
//   * an observation's factor row arrives as two 8-byte loads per lane (elements D-1-j and D-17-j: 128 contiguous bytes
//     per half and lane row) and its rank-1 update is 48 v_fmac_f64_dpp row_newbcast instructions -- lane i of the lane
//     row is the broadcast source of row i -- each doing the work of FOUR rows: 12 vector instructions per observation,
//     the cost of the three v_mfma_f64_16x16x4_f64 per four observations they replace (the two share the FP64 pipe);
//     blocks (0,0), (1,0), (1,1) are accumulated, block (0,1) is block (1,0)'s transpose (once per row, through LDS);
//   * observations past a piece's end are gathered from beyond the factor matrix with buffer loads, which return zero
//     there: no masks in the loop;
//   * a row of more than T observations is cut into 2 or 4 equal pieces on neighbouring lane rows of ONE wave, summed by
//     two butterfly steps over the lane rows; only a row of more than 4 T observations spans waves (equal parts of at
//     most 4 T, one reduced partial per wave through the slab, the part that arrives last sums them -- each lane row a
//     quarter of the slots, then the butterfly again -- and finishes the row);
//   * the LDL' factorisation with the forward solve riding along and the backward solve are dpp_rows32.h's: ~1,250
//     vector instructions per FOUR rows instead of ~800 per row;
//   * the host deals the ROUNDS (four lane rows' worth of work) to exactly as many waves as the row stream's CUs hold,
//     longest first to the least loaded wave: every wave carries the same cost, a launch is one balanced generation.
// How a row is cut depends on its own length and T only -- not on the launch, the shard or the number of GPUs -- so the
// sampled values do not depend on them either.
#include "bdf_common.h"
#include "dpp_rows32.h"
#include <algorithm>

#ifndef BDF_COL_WAVES
#define BDF_COL_WAVES 2            // waves per SIMD the kernel is compiled for (<= 256 registers)
#endif
#define COL_PSZ 800                // doubles per partial slot: 50 entries (A0[0..32], A1[16..32]) x 16 lanes

#ifdef BDF_K1_STAMPS      // diagnostic build: per wave {start, end, rounds, cycles by phase summed over the wave's rounds} (s_memtime; bdf_debug_stamps)
#define CSTAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc = 0, st_mid = 0, st_prior = 0, st_fin = 0, st_rng = 0, st_steps = 0, st_rounds = 0, st_parts = 0, st_fins = 0, st_folds = 0; const unsigned long long st_begin = st_t
#define CSTAMP(x) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); x += now_ - st_t; st_t = now_; } while (0)
#else
#define CSTAMP_DECL do { } while (0)
#define CSTAMP(x) do { } while (0)
#endif

namespace {

// K: the observation's position in its chunk of 16 (lane K of the lane row holds its value minus the mean)
template <int DR, int K>
__device__ __forceinline__ void col_obs(double (&A0)[33], double (&A1)[33], double v0, double v1, double r)
{
    col_rank1<DR, 0>(A0, A1, v0, v1);
    fm1_run<K>(A0[32], r, v0);
    fm1_run<K>(A1[32], r, v1);
}

__device__ __forceinline__ double col_ld(__amdgpu_buffer_rsrc_t rs, uint32_t off)
{
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u2v x = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
    return __hiloint2double((int)x.y, (int)x.x);
}

// the gathers of observations K0 .. K0 + 3 of the chunk (ids in idw, one per lane of the lane row)
template <bool FULL, int K0>
__device__ __forceinline__ void col_gather4(double (&b)[4][2], __amdgpu_buffer_rsrc_t rs, uint32_t idw, uint32_t rowb, uint32_t eo, bool ok1)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t id;
        if (q == 0) id = row_bcast_u32<K0>(idw); else if (q == 1) id = row_bcast_u32<K0 + 1>(idw);
        else if (q == 2) id = row_bcast_u32<K0 + 2>(idw); else id = row_bcast_u32<K0 + 3>(idw);
        // (idw holds the observations' ROW OFFSETS, id x row bytes, made once per chunk of sixteen: the broadcast and the addition of the
        // lane's element offset are then ONE v_add_u32_dpp per observation where broadcast, multiply and add were three instructions)
        if constexpr (FULL) {
            // eo: byte offset of element D-17-j; element D-1-j sits 128 bytes further
            const uint32_t base = id + eo;
            b[q][1] = col_ld(rs, base);
            b[q][0] = col_ld(rs, base + 128u);
        } else {
            // eo: byte offset of element D-1-j; the padded columns (D-17-j < 0) read beyond the matrix: zero
            const uint32_t base = id + eo;
            b[q][0] = col_ld(rs, base);
            b[q][1] = col_ld(rs, ok1 ? base - 128u : 0xffffffffu);
        }
    }
}
template <int DR, int K0>
__device__ __forceinline__ void col_compute4(double (&A0)[33], double (&A1)[33], const double (&b)[4][2], double r)
{
    col_obs<DR, K0>(A0, A1, b[0][0], b[0][1], r);
    col_obs<DR, K0 + 1>(A0, A1, b[1][0], b[1][1], r);
    col_obs<DR, K0 + 2>(A0, A1, b[2][0], b[2][1], r);
    col_obs<DR, K0 + 3>(A0, A1, b[3][0], b[3][1], r);
}

// butterfly over the lane rows: v += m * (v of lane ^ X), the 50 accumulated entries
template <int X>
__device__ __forceinline__ void col_fold(double (&A0)[33], double (&A1)[33], double m)
{
#pragma unroll
    for (int i = 0; i < 33; i++) A0[i] = fma(__shfl_xor(A0[i], X), m, A0[i]);
#pragma unroll
    for (int i = 16; i < 33; i++) A1[i] = fma(__shfl_xor(A1[i], X), m, A1[i]);
}

// entry e of a partial slot: A0[e] for e < 33, A1[e - 17] for 33 <= e < 50
// (src always points at a slot of the row -- a lane row without a slot of its own reads the first one again and multiplies it by 0:
// a conditional load would become a branch per entry)
template <int E0, int E1, bool ADD>
__device__ __forceinline__ void col_slot_load(double (&A0)[33], double (&A1)[33], const double *src, double m)
{
#pragma unroll
    for (int e = E0; e < E1; e++) {
        const double v = src[e * 16];
        double &d = e < 33 ? A0[e] : A1[e - 17];
        if (ADD) d = fma(v, m, d); else d = v * m;
    }
}

// the prior's image (the accumulator-layout image of the index-reversed Lambda: prior_pack / k_prior) for the lane's two columns,
// every load in flight at once -- a batch of sixteen at a time cost a round 12 k cycles, a quarter of its life.  SC: agent-scope
// loads (the pack was written by a hyperprior draw this launch did not wait for: past the CU's non-coherent L1).
// The prior's image (768 doubles at D <= 32) into the wave's LDS by LDS-DMA: six instructions of 64 lanes x 16 bytes, no vector
// registers, past the CU's L1 (sc1: the pack may be the hyperprior draw's of a moment ago).  Issued at the START of a round when
// the draw is already there -- it lands under the accumulation -- and read back with ds_read when the round adds the prior:
// 66 agent-scope loads in three dependent batches were 3.9 k cycles of every round (stamps), a twelfth of its life.
__device__ __forceinline__ void col_prior_dma(const double *image, unsigned lds_dst, int lane)
{
    const void *sbase = (const void *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)image >> 32)) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)image));
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const unsigned voff = (unsigned)k * 1024u + (unsigned)lane * 16u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst + (unsigned)k * 1024u), "s"(sbase) : "memory");
    }
}
template <int S, int I, int DR>
__device__ __forceinline__ void col_prior_lds(double (&A)[33], const double *img, double alpha, int j)
{
    if constexpr (I < DR) {
        A[I] = fma(alpha, A[I], img[sys_off<S, I>(j)]);
        col_prior_lds<S, I + 1, DR>(A, img, alpha, j);
    }
}

template <int DR, bool FULL>
__global__ __launch_bounds__(64, BDF_COL_WAVES) void k_rows_col(SampleArgs a_in, ColPlanDev p_in, uint32_t fac_bytes)
{
    __shared__ __attribute__((aligned(16))) double lds[4 * 272 + 768];      // block (1,0) of the four systems on its way to block (0,1) | the prior's image
    const int w = blockIdx.x;
    if (w >= p_in.n_waves) return;
    const int r_end = p_in.wave_round[w + 1];
    // (64 shards: two thousand waves on ONE word are served one after the other, ~12 ns each, and a wave's first loads queue behind
    // its own atomic -- measured: the launch 60 us instead of 39)
    if (a_in.span && threadIdx.x == 0) atomicMin(a_in.span + 2 * (w & 63), (unsigned long long)__builtin_amdgcn_s_memrealtime());
    CSTAMP_DECL;
#pragma nounroll
    for (int rd = p_in.wave_round[w]; rd < r_end; rd++) {
    // (the arguments and the lane through an index the compiler cannot see through: what it would hoist out of the loop -- the
    // arguments' fields, the constants of the normals' polynomials -- would stay in registers across the whole round)
    int zero = 0, lane = threadIdx.x;
    asm volatile("" : "+s"(zero), "+v"(lane));
    const SampleArgs &a = (&a_in)[zero];
    const ColPlanDev &p = (&p_in)[zero];
    const int j = lane & 15, g = lane >> 4;
    const TermDev &T = a.t[0];
    const int D = FULL ? 32 : a.D;
    const int ec0 = D - 1 - j, ec1 = D - 17 - j;          // natural index of the reversed elements j, 16 + j (ec1 < 0: padding)
    const int n0 = ec0, n1 = ec1 >= 0 ? ec1 : 0;
    const bool ok1 = ec1 >= 0;
    const uint32_t rowb = (uint32_t)D * 8u;
    const uint32_t eo = FULL ? (uint32_t)ec1 * 8u : (uint32_t)ec0 * 8u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)T.fac[0], 0, (int)fac_bytes, 0x00020000);
    const bool coded = T.packed != nullptr;                // (wave-uniform)
    const double mean = T.mean;
    {
        const ColJob jb = p.jobs[(int64_t)rd * 4 + g];
        const bool live = jb.row >= 0;
        const int n = live ? jb.count : 0;
        // the prior's image on its way into LDS under the accumulation, if the draw is there already (a launch that did not wait
        // for it: one look at its flag, no spinning here)
        const unsigned lds_img = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) double *)(lds + 4 * 272));
        bool have_prior = true;
        if (a.ready) have_prior = __builtin_amdgcn_readfirstlane((int)((int32_t)(__hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ready_want) >= 0)) != 0;
        if (have_prior) col_prior_dma(a.prior_c, lds_img, lane);
        // ---- accumulate: S~ (blocks (0,0), (1,0), (1,1)) and W r ----
        double A0[33], A1[33];
#pragma unroll
        for (int i = 0; i < 33; i++) { A0[i] = 0.0; A1[i] = 0.0; }
        int nmax = n;
        nmax = max(nmax, __shfl_xor(nmax, 16));
        nmax = max(nmax, __shfl_xor(nmax, 32));
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        if (nmax > 0) {
            const uint32_t *packed = coded ? T.packed + jb.q_begin : nullptr;
            const int32_t *colidx = T.colidx + jb.q_begin;
            const double *vals = T.vals + jb.q_begin;
            const double *table = T.table;
            // a chunk's ids and values minus the mean: lane j of the lane row takes observation c0 + j of its piece; positions
            // past the piece's end take an id beyond the factor matrix (gathers there return zero) and the value 0
#define COL_CHUNK(c0, IDW, R)                                                                     \
            {                                                                                     \
                const int o_ = (c0) + j;                                                          \
                IDW = 0xffffffu; R = 0.0;                                                         \
                if (o_ < n) {                                                                     \
                    if (coded) { const uint32_t pw_ = packed[o_]; IDW = pw_; R = table[pw_ >> 24] - mean; } \
                    else { IDW = (uint32_t)colidx[o_]; R = vals[o_] - mean; }                     \
                }                                                                                 \
                IDW = __umul24(IDW, rowb);      /* the row's byte offset (the low 24 bits of a packed word are the id) */ \
            }
            uint32_t idw_c, idw_n;
            double r_c, r_n;
            COL_CHUNK(0, idw_c, r_c)
            double ba[4][2], bb[4][2];
            col_gather4<FULL, 0>(ba, rs, idw_c, rowb, eo, ok1);
            for (int c0 = 0;; c0 += 16) {
                COL_CHUNK(c0 + 16, idw_n, r_n)
                col_gather4<FULL, 4>(bb, rs, idw_c, rowb, eo, ok1);
                __builtin_amdgcn_sched_barrier(0);
                col_compute4<DR, 0>(A0, A1, ba, r_c);
                if (c0 + 4 >= nmax) break;
                col_gather4<FULL, 8>(ba, rs, idw_c, rowb, eo, ok1);
                __builtin_amdgcn_sched_barrier(0);
                col_compute4<DR, 4>(A0, A1, bb, r_c);
                if (c0 + 8 >= nmax) break;
                col_gather4<FULL, 12>(bb, rs, idw_c, rowb, eo, ok1);
                __builtin_amdgcn_sched_barrier(0);
                col_compute4<DR, 8>(A0, A1, ba, r_c);
                if (c0 + 12 >= nmax) break;
                col_gather4<FULL, 0>(ba, rs, idw_n, rowb, eo, ok1);
                __builtin_amdgcn_sched_barrier(0);
                col_compute4<DR, 12>(A0, A1, bb, r_c);
                if (c0 + 16 >= nmax) break;
                idw_c = idw_n; r_c = r_n;
            }
#undef COL_CHUNK
        }
#ifdef BDF_K1_STAMPS
        asm volatile("s_nop 0" :: "v"(A0[0]), "v"(A0[32]), "v"(A1[31]));
        st_steps += (unsigned long long)nmax; st_rounds++;
#endif
        CSTAMP(st_acc);
        // ---- pieces of one row on neighbouring lane rows: butterfly sums (every lane row of the group ends with the row's sums) ----
        const unsigned fl = (unsigned)jb.flags;
        if (__builtin_amdgcn_readfirstlane((int)(__ballot((fl & COLF_PAIR) != 0) != 0ull))) {
#ifdef BDF_K1_STAMPS
            st_folds++;
#endif
            col_fold<16>(A0, A1, (fl & COLF_PAIR) ? 1.0 : 0.0);
            if (__builtin_amdgcn_readfirstlane((int)(__ballot((fl & COLF_QUAD) != 0) != 0ull)))
                col_fold<32>(A0, A1, (fl & COLF_QUAD) ? 1.0 : 0.0);
        }
        // ---- a row that spans waves: this part's sums to the slab; the part that arrives last sums them all and keeps the row ----
        if (__builtin_amdgcn_readfirstlane((int)(fl & COLF_MULTI))) {
            const int srow = __builtin_amdgcn_readfirstlane(jb.srow);
            double *dst = p.partials + (int64_t)__builtin_amdgcn_readfirstlane(jb.slot) * COL_PSZ + j;
            if (g == 0) {
                // write-through (sc1) stores: the slab needs no L2 write-back (k_rows' protocol)
#pragma unroll
                for (int e = 0; e < 33; e++) __hip_atomic_store(dst + e * 16, A0[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int e = 33; e < 50; e++) __hip_atomic_store(dst + e * 16, A1[e - 17], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const ColSplit sr = p.rows[srow];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(p.arrived + srow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
#ifdef BDF_K1_STAMPS
            st_parts++;
            if (old == sr.n_slots - 1) st_fins++;
#endif
            if (old != sr.n_slots - 1) { CSTAMP(st_mid); continue; }                    // not the row's last part
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) p.arrived[srow] = 0;                     // ready for the next launch
            // lane row g sums slots g, g + 4, ... in that order (the first one straight into the registers: all its loads in
            // flight at once), then the butterfly: the same sums whichever part does it
            const double *src = p.partials + (int64_t)sr.slot_begin * COL_PSZ + j;
            col_slot_load<0, 50, false>(A0, A1, src + (int64_t)(g < sr.n_slots ? g : 0) * COL_PSZ, g < sr.n_slots ? 1.0 : 0.0);
            for (int s0 = 4; s0 < sr.n_slots; s0 += 4) {
                const bool on = s0 + g < sr.n_slots;
                const double *sp = src + (int64_t)(on ? s0 + g : 0) * COL_PSZ;
                const double m = on ? 1.0 : 0.0;
                col_slot_load<0, 16, true>(A0, A1, sp, m);
                asm volatile("" ::: "memory");
                col_slot_load<16, 33, true>(A0, A1, sp, m);
                asm volatile("" ::: "memory");
                col_slot_load<33, 50, true>(A0, A1, sp, m);
                asm volatile("" ::: "memory");
            }
            col_fold<16>(A0, A1, 1.0);
            col_fold<32>(A0, A1, 1.0);
        }
        CSTAMP(st_mid);
        // ---- normals (after the sums: a part of a spanning row that is not its last draws none): lane p of the lane row draws pair p of the row's stream; column c wants number D - 1 - c ----
        double z0 = 0.0, z1 = 0.0;
        {
            double ze = 0.0, zo = 0.0;
            if (live && 2 * j < D) bdf_normal_pair(a.seed, a.sweep, BDF_P_ROW, a.entity_tag, (uint64_t)(uint32_t)jb.orig, (uint32_t)j, ze, zo);
            const int base = lane & 48;
            const double ze0 = __shfl(ze, base + (n0 >> 1)), zo0 = __shfl(zo, base + (n0 >> 1));
            const double ze1 = __shfl(ze, base + (n1 >> 1)), zo1 = __shfl(zo, base + (n1 >> 1));
            z0 = (n0 & 1) ? zo0 : ze0;
            z1 = ok1 ? ((n1 & 1) ? zo1 : ze1) : 0.0;
        }
        asm volatile("" ::: "memory");
        CSTAMP(st_rng);
        // ---- block (0,1) = block (1,0)': lane j's entry (i, 16 + j) is lane i's entry (16 + j, i) ----
        {
            double *tl = lds + g * 272;
#pragma unroll
            for (int r = 0; r < 16; r++) tl[r * 17 + j] = A0[16 + r];
            wave_sync();
#pragma unroll
            for (int i = 0; i < 16; i++) A1[i] = tl[j * 17 + i];
            wave_sync();
        }
        CSTAMP(st_mid);
        // ---- P~ = alpha S~ + Lambda~, b~ = alpha W r + Lambda mu: the prior pack (polled for when the launch did not wait for the draw) ----
        const double alpha = term_alpha(T);
        if (!have_prior) {
            int spins = 0;
            while ((int32_t)(__hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ready_want) < 0) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1 << 22)) { if (lane == 0) atomicOr_system(a.flag, 16); break; }
            }
            col_prior_dma(a.prior_c, lds_img, lane);
        }
        {
            const double *pb = a.prior_b + (a.mu_is_matrix && live ? (int64_t)jb.row * D : 0);
            const double b0 = __hip_atomic_load(pb + n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double b1 = __hip_atomic_load(pb + n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the image has landed (nothing else orders a ds_read behind an LDS-DMA)
            wave_sync();
            const double *img = lds + 4 * 272;
            col_prior_lds<0, 0, DR>(A0, img, alpha, j);
            col_prior_lds<1, 0, DR>(A1, img, alpha, j);
            A0[32] = fma(alpha, A0[32], b0);
            A1[32] = ok1 ? fma(alpha, A1[32], b1) : 0.0;
            wave_sync();                                           // (the next round's image goes to the same place)
        }
#ifdef BDF_K1_STAMPS
        asm volatile("s_nop 0" :: "v"(A0[0]), "v"(A0[31]), "v"(A1[0]), "v"(A1[31]));
#endif
        CSTAMP(st_prior);
        // ---- LDL' with the forward solve riding along, backward solve, the draw ----
        double d0 = 1.0, d1 = 1.0;
        fin_factor<DR, 0>(A0, A1, d0, d1, j);
        const bool lead = live && (fl & COLF_LEADER);
        if (lead && (!(d0 > 0.0) || (ok1 && !(d1 > 0.0)))) atomicOr_system(a.flag, 1);      // not positive definite
        const double rd0 = fast_rcp(d0), rd1 = fast_rcp(d1);
        double y0 = fma(z0, fast_rsqrt(d0), A0[32] * rd0), y1 = fma(z1, fast_rsqrt(d1), A1[32] * rd1);
        fin_backward<DR - 1>(A0, A1, y0, y1, rd0, rd1, j);
        // (write-through: with a.done the hyperprior's sums read the rows while this launch is still running, from CUs behind other L2s)
        if (lead) __hip_atomic_store(a.out + ((int64_t)jb.row * D + ec0), y0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lead && ok1) __hip_atomic_store(a.out + ((int64_t)jb.row * D + ec1), y1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CSTAMP(st_fin);
    }
    }
    if (a_in.done) {
        // this wave's rows are in memory (write-through stores, drained): one more wave of the launch done -- what the entity's
        // hyperprior chain polls for (k_hyper_chain) instead of waiting for the launch's completion event
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a_in.done + BDF_DONE_STRIDE * (w & (BDF_DONE_SHARDS - 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a_in.span && threadIdx.x == 0) atomicMax(a_in.span + 2 * (w & 63) + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#ifdef BDF_K1_STAMPS
    if (threadIdx.x == 0 && a_in.b_dump && w < 65536) {
        unsigned long long *d = (unsigned long long *)a_in.b_dump + (size_t)w * 16;
        d[0] = st_begin; d[8] = st_t; d[6] = st_rounds; d[7] = st_steps;
        d[1] = st_parts; d[2] = st_fins; d[3] = st_folds;
        d[11] = st_rng; d[12] = st_acc; d[13] = st_mid; d[14] = st_prior; d[15] = st_fin;
        d[9] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4));
        d[10] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    }
#endif
}

// ---- host: rows -> units (a row, or its 2 / 4 pieces, or the parts of a row that spans waves) -> rounds -> waves ------------
#define COL_C_OBS 56.0             // cost model (vector instructions): per observation of a round's longest piece,
#define COL_C_FIN 1460.0           // the finish of a round (normals, prior, factorisation, solves),
#define COL_C_FOLD 180.0           // the butterfly sums of a round with pieces,
#define COL_C_PART 700.0           // a part's store to the slab

struct Round { ColJob job[4]; double cost; };

ColJob idle_job() { return ColJob{-1, 0, 0, 0, -1, 0, 0}; }

}  // namespace

int bdf_col_plan_build(bdf_ctx *ctx, const std::vector<bdf_row_ref> &rows, int T, int64_t slots, bdf_col_plan &plan)
{
    std::vector<Round> rounds;
    std::vector<ColSplit> splits;
    struct Unit { int32_t row, orig; int64_t qb; int32_t n; };
    std::vector<Unit> singles, pairs;
    int32_t n_slots_total = 0;
    auto quad_round = [&](const bdf_row_ref &rr, int64_t b0, int64_t b1, int32_t srow, int32_t slot, double fin_share) {
        Round R;
        const int64_t m = b1 - b0;
        int32_t lmax = 0;
        for (int q = 0; q < 4; q++) {
            const int64_t p0 = b0 + m * q / 4, p1 = b0 + m * (q + 1) / 4;
            R.job[q] = ColJob{rr.out, rr.orig, rr.qb + p0, (int32_t)(p1 - p0), srow, slot,
                              (int32_t)(COLF_PAIR | COLF_QUAD | (q == 0 ? COLF_LEADER : 0) | (srow >= 0 ? COLF_MULTI : 0))};
            lmax = std::max(lmax, (int32_t)(p1 - p0));
        }
        // (a part of a spanning row is ranked ABOVE what the model gives it: the part that arrives last pays the whole finish, the slab's
        // round trips and the slots' sums, and as the YOUNGER wave of its SIMD -- 900 cycles per observation step against the older
        // wave's 430 -- it was the launch's tail; ranked among the heavy rounds it is dispatched first and runs as the older one:
        // users' launch alone 31.9 -> 29.9 us, movies' 33.9 -> 32.9, the iteration +2.5 %: profiles/r05_k1c_part_cost.txt)
        static const double part_extra = getenv("BDF_COL_PART_COST") ? atof(getenv("BDF_COL_PART_COST")) : 3000.0;
        R.cost = COL_C_OBS * lmax + 2 * COL_C_FOLD + (srow >= 0 ? COL_C_PART + fin_share + part_extra : COL_C_FIN);
        rounds.push_back(R);
    };
    for (const bdf_row_ref &rr : rows) {
        const int64_t n = rr.cnt;
        if (n <= T) singles.push_back(Unit{rr.out, rr.orig, rr.qb, (int32_t)n});
        else if (n <= 2 * (int64_t)T) pairs.push_back(Unit{rr.out, rr.orig, rr.qb, (int32_t)n});
        else if (n <= 4 * (int64_t)T) quad_round(rr, 0, n, -1, 0, 0.0);
        else {
            const int64_t W = (n + 4 * (int64_t)T - 1) / (4 * (int64_t)T);
            BDF_REQUIRE(W < (1 << 20), BDF_ERR_ARG, "bdf_sample_rows: a row of %lld observations", (long long)n);
            const int32_t srow = (int32_t)splits.size();
            splits.push_back(ColSplit{n_slots_total, (int32_t)W});
            // (the part that arrives last pays the finish and the slots' sums: on average a share each)
            const double share = (COL_C_FIN + 2 * COL_C_FOLD + 60.0 * (double)((W + 3) / 4)) / (double)W;
            for (int64_t q = 0; q < W; q++) quad_round(rr, n * q / W, n * (q + 1) / W, srow, n_slots_total + (int32_t)q, share);
            n_slots_total += (int32_t)W;
        }
    }
    // pairs two to a round, singles four to a round, like with like (the round lasts as long as its longest piece)
    auto by_len = [](const Unit &x, const Unit &y) { return x.n > y.n; };
    std::stable_sort(pairs.begin(), pairs.end(), by_len);
    std::stable_sort(singles.begin(), singles.end(), by_len);
    auto pair_jobs = [](const Unit &u, ColJob *dst) {
        const int32_t h = u.n / 2;
        dst[0] = ColJob{u.row, u.orig, u.qb, h, -1, 0, COLF_PAIR | COLF_LEADER};
        dst[1] = ColJob{u.row, u.orig, u.qb + h, u.n - h, -1, 0, COLF_PAIR};
    };
    size_t si = 0;
    for (size_t pi = 0; pi < pairs.size(); pi += 2) {
        Round R;
        pair_jobs(pairs[pi], R.job);
        int32_t lmax = pairs[pi].n - pairs[pi].n / 2;
        if (pi + 1 < pairs.size()) pair_jobs(pairs[pi + 1], R.job + 2);
        else {
            // the odd pair out shares its round with the two longest rows left
            for (int q = 2; q < 4; q++) {
                if (si < singles.size()) {
                    const Unit &u = singles[si++];
                    R.job[q] = ColJob{u.row, u.orig, u.qb, u.n, -1, 0, COLF_LEADER};
                    lmax = std::max(lmax, u.n);
                } else R.job[q] = idle_job();
            }
        }
        R.cost = COL_C_OBS * lmax + COL_C_FOLD + COL_C_FIN;
        rounds.push_back(R);
    }
    for (; si < singles.size(); si += 4) {
        Round R;
        for (int q = 0; q < 4; q++) {
            if (si + q < singles.size()) {
                const Unit &u = singles[si + q];
                R.job[q] = ColJob{u.row, u.orig, u.qb, u.n, -1, 0, COLF_LEADER};
            } else R.job[q] = idle_job();
        }
        R.cost = COL_C_OBS * singles[si].n + COL_C_FIN;
        rounds.push_back(R);
    }
    plan.n_rounds = (int64_t)rounds.size();
    plan.n_waves = 0;
    plan.n_split_rows = (int32_t)splits.size();
    if (rounds.empty()) return BDF_OK;
    // ONE ROUND PER WAVE.  Two waves share a SIMD (250 registers each), and the older of the two takes the issue slots it wants: a
    // heavy round beside a medium one runs at a lone wave's pace while its partner crawls (stamps: 430 against 900 cycles per
    // observation step).  The dispatcher deals single-wave workgroups round-robin -- wave w and wave w + (SIMDs) meet on one SIMD
    // (1,024 apart on the whole chip, 896 in the shader engines a CU mask has taken a CU from) -- so the ORDER of the waves decides
    // who shares: costliest first paired the heaviest round with the median one and the median with the lightest (SIMD totals of
    // 194 against 76 observation steps on MovieLens; the launch ended with the first kind).  Here the first S2 waves hold the
    // heavier rounds and the next S2 bring the lighter ones LIGHTEST FIRST, so that wave w's partner is its complement and every
    // SIMD gets about the same sum.  S2 = 896, the SHORTER of the two distances, whatever the stream's CUs (the order must not
    // depend on them): with 1,024 the waves 896 .. 1,023 -- still heavy -- became the partners of the heaviest ones in the masked
    // shader engines and the launch took 41 us instead of 37 on the row stream's 248 CUs; with 896 a partner is at worst 128
    // places -- a few observation steps -- from the exact complement, under either mask: 33.1 / 34.9 us (users' / movies' launch
    // alone, all CUs; 33.2 / 35.2 on 248) against 35.4 / 37.9 costliest-first (profiles/r05_k1c_wave_order.txt).  With fewer rounds
    // than 2 S2 the heaviest run alone; rounds beyond 2 S2 follow costliest first and take whichever slot comes free.
    (void)slots;
    const int64_t nw = (int64_t)rounds.size();
    std::vector<int32_t> rank(rounds.size());
    for (size_t i = 0; i < rank.size(); i++) rank[i] = (int32_t)i;
    std::stable_sort(rank.begin(), rank.end(), [&](int32_t x, int32_t y) { return rounds[(size_t)x].cost > rounds[(size_t)y].cost; });
    static const int64_t S2 = getenv("BDF_COL_PAIR_DISTANCE") ? std::max<int64_t>(0, atoll(getenv("BDF_COL_PAIR_DISTANCE"))) : 896;
    std::vector<int32_t> idx;
    idx.reserve(rounds.size());
    if (S2 == 0 || nw <= S2) idx = rank;                       // (0: the plain costliest-first order)
    else {
        const int64_t two = std::min<int64_t>(nw, 2 * S2);      // the rounds of the first two generations: ranks 0 .. two - 1
        const int64_t P = two - S2, A = S2 - P;                 // P pairs; the A heaviest alone
        for (int64_t i = 0; i < P; i++) idx.push_back(rank[(size_t)(A + i)]);
        for (int64_t i = 0; i < A; i++) idx.push_back(rank[(size_t)i]);
        for (int64_t i = 0; i < P; i++) idx.push_back(rank[(size_t)(two - 1 - i)]);
        for (int64_t i = two; i < nw; i++) idx.push_back(rank[(size_t)i]);
    }
    std::vector<ColJob> jobs;
    std::vector<int32_t> wave_round;
    jobs.reserve(rounds.size() * 4);
    wave_round.push_back(0);
    for (int32_t i : idx) {
        for (int q = 0; q < 4; q++) jobs.push_back(rounds[(size_t)i].job[q]);
        wave_round.push_back((int32_t)(jobs.size() / 4));
    }
    plan.cost_max = rounds[(size_t)rank.front()].cost;
    plan.cost_min = rounds[(size_t)rank.back()].cost;
    plan.n_waves = (int32_t)nw;
    BDF_HIP(hipMalloc((void **)&plan.jobs_dev, jobs.size() * sizeof(ColJob)));
    BDF_HIP(hipMemcpy(plan.jobs_dev, jobs.data(), jobs.size() * sizeof(ColJob), hipMemcpyHostToDevice));
    BDF_HIP(hipMalloc((void **)&plan.wave_round_dev, wave_round.size() * sizeof(int32_t)));
    BDF_HIP(hipMemcpy(plan.wave_round_dev, wave_round.data(), wave_round.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    BDF_HIP(hipMalloc((void **)&plan.rows_dev, std::max<size_t>(splits.size() * sizeof(ColSplit), 8)));
    if (!splits.empty()) BDF_HIP(hipMemcpy(plan.rows_dev, splits.data(), splits.size() * sizeof(ColSplit), hipMemcpyHostToDevice));
    BDF_HIP(hipMalloc((void **)&plan.partials_dev, std::max<size_t>((size_t)n_slots_total * COL_PSZ * sizeof(double), 8)));
    BDF_HIP(hipMalloc((void **)&plan.arrived_dev, std::max<size_t>(splits.size() * sizeof(int32_t), 8)));
    // (on the launch stream: a memset on the NULL stream can run under the first launch -- DESIGN.md section 8)
    BDF_HIP(hipMemsetAsync(plan.arrived_dev, 0, std::max<size_t>(splits.size() * sizeof(int32_t), 8), ctx->stream));
    return BDF_OK;
}

void bdf_col_plan_free(bdf_col_plan &plan)
{
    if (plan.jobs_dev) (void)hipFree(plan.jobs_dev);
    if (plan.wave_round_dev) (void)hipFree(plan.wave_round_dev);
    if (plan.rows_dev) (void)hipFree(plan.rows_dev);
    if (plan.partials_dev) (void)hipFree(plan.partials_dev);
    if (plan.arrived_dev) (void)hipFree(plan.arrived_dev);
    plan = bdf_col_plan{};
}

int bdf_col_launch(bdf_ctx *ctx, const SampleArgs &a, const bdf_col_plan &plan, int64_t M_other, hipEvent_t e0, hipEvent_t e1)
{
    if (plan.n_waves <= 0) return BDF_OK;
    ColPlanDev p;
    p.jobs = plan.jobs_dev; p.wave_round = plan.wave_round_dev; p.n_waves = plan.n_waves; p._pad = 0;
    p.rows = plan.rows_dev; p.partials = plan.partials_dev; p.arrived = plan.arrived_dev;
    const int64_t bytes = M_other * (int64_t)a.D * 8;
    BDF_REQUIRE(bytes < ((int64_t)1 << 32), BDF_ERR_ARG, "bdf_sample_rows: the column kernel needs a factor matrix below 4 GiB");
    const uint32_t fb = (uint32_t)bytes;
    const dim3 grid((unsigned)plan.n_waves), block(64);
    const int DR = (a.D + 3) / 4 * 4;
#define COL_LAUNCH(DRV, FULLV) hipExtLaunchKernelGGL((k_rows_col<DRV, FULLV>), grid, block, 0, ctx->stream, e0, e1, 0, a, p, fb)
    if (a.D == 32) COL_LAUNCH(32, true);
    else if (DR <= 20) COL_LAUNCH(20, false);
    else if (DR <= 24) COL_LAUNCH(24, false);
    else if (DR <= 28) COL_LAUNCH(28, false);
    else COL_LAUNCH(32, false);
#undef COL_LAUNCH
    BDF_HIP(hipGetLastError());
    return BDF_OK;
}
