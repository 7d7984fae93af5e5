/*
 * bdf.h -- C ABI of libbdf_hip.so: the MI355X (gfx950) implementation of the Gibbs-sweep
 * hot path of BayesianDataFusion.jl (latent-row sampler + hyperprior + side-information
 * beta update + test-set prediction).
 *
 * The reference has no FFI for this path (pure Julia, multiple dispatch); the entry points
 * below are the seams a maintainer would `ccall` from the reference's own functions.  Each
 * declaration cites the reference interface (file:line under the reference tree) it replaces.
 * INTEGRATION.md shows the Julia-side binding.
 *
 * Conventions
 *   - every function returns BDF_OK (0) or a negative BDF_ERR_* code; bdf_last_error() gives
 *     the message (the reference throws ArgumentError / DimensionMismatch / BoundsError).
 *   - no exceptions, no C++ types, no torch types cross this boundary.
 *   - "dev" pointers are device (HBM) addresses on the context's GPU; "host" pointers are
 *     caller-owned host memory valid for the duration of the call only.
 *   - matrices are column-major as in Julia: an entity's sample matrix is D x N (one column
 *     of D doubles per entity instance), beta is numF x D, a dense F is N x numF.
 *   - ids crossing the boundary from the reference's data model (relation / test pairs) are
 *     1-based like the DataFrame holds them; row lists and ranges are 0-based.
 *   - one host thread per context (macau.jl runs the Gibbs loop on one task); all work is
 *     enqueued on the context's HIP stream, asynchronously unless stated.
 */
#ifndef BDF_H
#define BDF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BDF_OK           0
#define BDF_ERR_ARG     (-1)  /* ArgumentError / DimensionMismatch in the reference */
#define BDF_ERR_BOUNDS  (-2)  /* BoundsError: id outside 1..dims                     */
#define BDF_ERR_HIP     (-3)  /* HIP runtime error                                  */
#define BDF_ERR_NOTPD   (-4)  /* a matrix that must be positive definite is not     */
#define BDF_ERR_NOGPU   (-5)  /* no usable gfx950 device                            */

#define BDF_MAX_MODES 4       /* modes per relation (matrix = 2, tensors up to 4)  */
#define BDF_MAX_TERMS 4       /* relations summed per entity row                   */
#define BDF_MAX_D     64      /* num_latent                                        */

/* RNG stream purposes (DESIGN.md "RNG contract"); Philox4x32-10 counters are
 * (row, pair | row_hi, sweep, purpose<<24 | entity_tag), key = seed. */
#define BDF_P_ROW       1
#define BDF_P_BETA_E1   2
#define BDF_P_BETA_E2   3
#define BDF_P_NW_NORMAL 4
#define BDF_P_GAMMA_N   5
#define BDF_P_GAMMA_U   6
#define BDF_P_NW_MEAN   7
#define BDF_P_BETA_REL1 8   /* sample_beta_rel: noise per observation (row = observation)  */
#define BDF_P_BETA_REL2 9   /* sample_beta_rel: noise per feature    (row = feature)        */

typedef struct bdf_ctx   bdf_ctx;    /* device, stream, seed, sweep counter, scratch        */
typedef struct bdf_rel   bdf_rel;    /* Relation.data :: IndexedDF / FastIDF on the device  */
typedef struct bdf_pairs bdf_pairs;  /* Relation.test_vec (+ running prediction state)      */
typedef struct bdf_feat  bdf_feat;   /* Entity.F operator (dense / CSR / binary CSR / COO)  */
typedef struct bdf_comm  bdf_comm;   /* the ranks (GPUs) that share the entities' rows       */
typedef struct bdf_gibbs bdf_gibbs;  /* a whole Gibbs iteration enqueued from native code    */

const char *bdf_last_error(void);
int bdf_version(void);

/* ---- context ------------------------------------------------------------------------ */
/* stream: a hipStream_t the caller owns (e.g. torch's current stream); NULL is the device's
 * default stream.  seed keys every random draw of the context. */
int bdf_ctx_create(int device, void *stream, uint64_t seed, bdf_ctx **out);
int bdf_ctx_destroy(bdf_ctx *ctx);
/* Gibbs iteration number (macau.jl:80 loop variable): part of every random-stream address; handed to
 * the launches that follow by value. */
int bdf_ctx_set_sweep(bdf_ctx *ctx, uint32_t sweep);
int bdf_ctx_advance_sweep(bdf_ctx *ctx);
/* A context on ANOTHER stream of main_ctx's device (created and owned by the library; bdf_ctx_destroy frees it), chosen so that
 * kernels on it really run beside those of main_ctx and of the contexts in `apart`: HIP multiplexes streams onto a few
 * hardware queues, and two streams that share one serialise each other (measured: 171 instead of 125 us per sweep).  A
 * timing test (two 60 us spins, together) picks among a few candidate streams; with no passing candidate the last one is
 * returned (correct, only slower).  The hyperprior of entity j is enqueued on such a context beside the rows of entity j+1. */
int bdf_ctx_create_side(bdf_ctx *main_ctx, bdf_ctx *const *apart, int n_apart, int reserved, bdf_ctx **out);
/* A row context on a stream of its own that leaves `reserve_cus` CUs (0, 8, 16, ...: whole CUs per XCD) free of its kernels
 * (hipExtStreamCreateWithCUMask); a side context created from it with reserved = 1 runs on exactly those CUs, with reserved =
 * 0 on the others.  The row sampler fills every CU it may use for the whole launch (its waves hold 468 of a SIMD's 512
 * registers): the hyperprior's small kernels, enqueued beside it, otherwise wait for slots -- measured, a one-workgroup kernel
 * beside a chip-filling one: 194 us on plain streams, 6 - 11 us on its own 16 / 8 CUs (tools/cu_mask_probe.hip).  Worth it
 * when the side work is small (MovieLens-sized entities); with reserve_cus = 0 all streams use the whole chip. */
int bdf_ctx_create_rows(int device, uint64_t seed, int reserve_cus, bdf_ctx **out);
/* Measurement support: HIP events (timing enabled) and "attach this pair to the next bdf_sample_rows launch of ctx":
 * the events ride on the row kernel's own dispatch packet (hipExtLaunchKernelGGL), so start/stop are the kernel's begin and
 * end on its stream without marker packets around it (an event pair recorded around a launch costs the stream ~6 us and
 * is counted into the interval).  bdf_event_elapsed_us waits for `stop`.  Either event may be NULL. */
int bdf_event_create(void **ev);
int bdf_event_destroy(void *ev);
int bdf_event_elapsed_us(void *start, void *stop, double *us);
int bdf_ctx_time_next_rows(bdf_ctx *ctx, void *start, void *stop);
/* the same for the hyperprior chain of ctx: `start` rides on the next bdf_hyper_sums' first kernel, `stop` on the next
 * bdf_hyper_sample's kernel */
int bdf_ctx_time_next_hyper(bdf_ctx *ctx, void *start, void *stop);
int bdf_ctx_sync(bdf_ctx *ctx);   /* waits for the stream; BDF_ERR_NOTPD if a kernel met a non-positive-definite matrix */
/* Conditions that are not errors in the reference either, accumulated by bdf_ctx_sync and returned (and cleared) here:
 * BDF_WARN_CG_MAXITER -- a conjugate-gradient column of the beta solve was still above its tolerance after maxiter iterations
 * (cg_AtA, src/parallel_cg.jl:73-93, returns such a column as it stands, silently; hosts may want to say so). */
#define BDF_WARN_CG_MAXITER 64u
int bdf_ctx_warnings(bdf_ctx *ctx, uint32_t *bits_out);
/* tuning: observations per K1 work item (rows with more are split over several wavefronts; default 192), and the size
 * of the pieces such a row is split into (default 128; set_item_size resets it to 2/3 of the item size).  Until either is set
 * (and again after set_item_size(ctx, 0)) the sizes are automatic: the defaults, and up to 2048 / 1365 for launches with
 * hundreds of waves per resident slot, where pieces only cost partial sums.  Results do not depend on them beyond the order
 * of the floating-point sums. */
int bdf_ctx_set_item_size(bdf_ctx *ctx, int observations);
int bdf_ctx_set_piece_size(bdf_ctx *ctx, int observations);
/* D <= 16, an entity of one two-mode relation: rows of at most max_observations observations are sampled FOUR TO A WAVE (16 lanes
 * and a column-per-lane 16 x 16 system each) by a launch of their own when the entity has at least min_rows rows -- at such
 * D the wave-per-row kernel is bound by its per-row instruction overhead.  Defaults 48 and 8192 (environment BDF_K1_SMALL,
 * BDF_K1_SMALL_MIN_ROWS); max_observations 0 turns it off.  Same sample up to the order of the floating-point sums. */
int bdf_ctx_set_small_rows(bdf_ctx *ctx, int max_observations, int64_t min_rows);
/* D > 16, an entity of one two-mode relation (shared or per-row prior means): rows of at most max_observations observations (at most
 * 16 at num_latent <= 32, 32 above -- rows of 17 .. 32 observations two to a lane, k_rows_lr32; -1 = num_latent / 2 up to that, the
 * default; 0 = off; environment BDF_LOWRANK) are drawn by the LOW-RANK SAMPLER
 * (k_rows_lr.hip) when a launch has at least min_rows of them (default 8192, BDF_LOWRANK_MIN_ROWS) and at least half as many as
 * the opposite entity has rows (min_rows = 0: whenever there is such a row).  It replaces sample_user_basic (src/sampling.jl:200-212) for those rows by another map from
 * standard normals to the SAME conditional distribution N(inv(P_i) b_i, inv(P_i)): D + n normals of the row's stream and an
 * n x n solve instead of a D x D inverse and factorisation (P_i = Lambda + rank n).  Sampled VALUES therefore differ from the
 * reference's map for those rows (the distribution does not: oracle/bdf_oracle.c orc_sample_row_lowrank is the same function,
 * proved equal in mean and covariance to inv(P_i) b_i, inv(P_i)); max_observations = 0 restores the reference's map for
 * every row. */
int bdf_ctx_set_lowrank(bdf_ctx *ctx, int max_observations, int64_t min_rows);
/* 16 < D <= 32, an entity of ONE two-mode relation without per-observation baselines (shared or per-row prior means): its rows are
 * sampled FOUR TO A WAVE in a column-per-lane layout from the first observation to the draw (k_rows_col.hip, "K1c": 16 lanes and
 * two columns per lane for each 32 x 32 system; sample_user_basic, src/sampling.jl:200-212 -- the SAME map from the row's
 * normals to the sample as the wave-per-row kernel, equal to rounding: only the order of the floating-point sums differs).  A row
 * of more than max_piece observations is cut into 2 or 4 equal pieces on neighbouring lane rows of one wave; a row of more than
 * 4 max_piece observations spans waves.  The cut depends on the row's own length and max_piece only, so the values do not depend
 * on the launch, the shard or the number of GPUs.  Default 128 (environment BDF_K1_COL: 0 = off, n = that piece size) for
 * launches whose item size the caller has not set (bdf_ctx_set_item_size keeps the wave-per-row kernel); a call here with
 * 8..4096 applies to every launch of the context, 0 turns it off, -1 restores the default. */
int bdf_ctx_set_col_rows(bdf_ctx *ctx, int max_piece);
/* report: how the most recent bdf_sample_rows launch of the entity with this entity_tag (all its chunks / shards since the tag's
 * previous iteration number) was dispatched on this context -- out[0] rows drawn by the low-rank sampler (k_rows_lr.hip, a
 * different map from the normals than sample_user_basic's, src/sampling.jl:200-212), out[1] rows by k_rows_small (D <= 16),
 * out[2] rows by k_rows_col (K1c), out[3] rows by the wave-per-row kernel k_rows, out[4] its work items (pieces included),
 * out[5] K1c's waves.  BDF_ERR_ARG if the context has launched no rows under that tag. */
int bdf_ctx_rows_dispatch(const bdf_ctx *ctx, uint32_t entity_tag, int64_t out[6]);
/* measurement: slot (dev, 64 pairs of uint64, each set to {~0, 0} by the caller) receives per pair s {earliest start, latest end}
 * of the waves w = s mod 64 of the NEXT K1c launch of the context, in ticks of the 100 MHz clock the XCDs share (s_memrealtime; one
 * atomic min / max per wave, sharded: one word would serialise two thousand waves): min / max over the pairs = the launch's
 * duration with no event packets around it.  bdf_gibbs_span_rows: the same for the next launch of an entity inside
 * bdf_gibbs_sweep. */
int bdf_ctx_span_next_rows(bdf_ctx *ctx, void *slot_dev);
/* parity hook: which gather path the row kernel takes.  0 = chosen by the sizes (default; env BDF_GATHER=general|wide sets the
 * initial value), 1 = the general path (any number of modes, per-observation baselines), 2 = the lean path with 64-bit row
 * offsets (num_latent > 32; what a factor matrix of 4 GiB or more needs, e.g. 10M rows at D = 64).  Same values on every path. */
int bdf_ctx_set_gather(bdf_ctx *ctx, int mode);
/* device memory for hosts without an allocator of their own (Julia); torch hosts pass tensors */
int bdf_dev_alloc(bdf_ctx *ctx, size_t bytes, void **dptr);
int bdf_dev_free(bdf_ctx *ctx, void *dptr);
int bdf_h2d(bdf_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int bdf_d2h(bdf_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);  /* synchronises */

/* ---- a1: IndexedDF / FastIDF  (src/IndexedDF.jl:10-21, 46-70) ----------------------- */
/* Host-only (needs no GPU): the IndexedDF constructor's index (IndexedDF.jl:10-19).  ids: nnz x n_modes
 * column-major, 1-based, id_bytes 4|8.  rowptr[m]: caller array of dims[m]+1 (0-based offsets);
 * rowids[m]: caller array of nnz (1-based COO row numbers, in original row order).
 * Errors: BDF_ERR_BOUNDS for an id outside 1..dims[m]. */
int bdf_index_build(int n_modes, const int64_t *dims, int64_t nnz, const void *ids, int id_bytes,
                    int64_t *const *rowptr, int64_t *const *rowids);
/* ids: nnz x n_modes column-major, 1-based, id_bytes = 4 (Int32) or 8 (Int64); values: nnz.
 * Builds, per mode, the adjacency index in ORIGINAL COO order (bit-exact with
 * IndexedDF.index) and its device CSR.  Errors: BDF_ERR_BOUNDS for an id outside 1..dims. */
/* (A two-mode relation whose values are at most 256 distinct numbers -- ratings -- is also kept as 8-bit value codes packed
 * with the other mode's id, 4 bytes per observation and mode: the row kernel's coded variant, bit-identical results.) */
int bdf_relation_create(bdf_ctx *ctx, int n_modes, const int64_t *dims, int64_t nnz,
                        const void *ids, int id_bytes, const double *values, bdf_rel **out);
int bdf_relation_destroy(bdf_rel *rel);
/* host view of index[mode]: rowptr[dims[mode]+1] (0-based offsets) and rowids[nnz] (1-based
 * COO row numbers), for getData/getCount/getI (IndexedDF.jl:41-43, 67-70) */
int bdf_relation_index(const bdf_rel *rel, int mode, const int64_t **rowptr, const int64_t **rowids);
/* valueMean (IndexedDF.jl:26) */
int bdf_relation_value_mean(const bdf_rel *rel, double *mean);
/* host copy of the degree-descending launch order of `mode` (0-based row numbers, n = dims[mode]) */
int bdf_relation_order(const bdf_rel *rel, int mode, int32_t *order_host);

/* ---- a3-a7: latent rows ---------------------------------------------------------------
 * One relation's contribution to the rows of the entity being sampled:
 * sample_user_basic (src/sampling.jl:200-212 matrix, :215-234 tensor) and the per-relation
 * body of sample_user2 (src/sampling.jl:270-283). */
typedef struct {
    const bdf_rel *rel;
    int32_t mode;                 /* 0-based mode of the sampled entity in rel              */
    int32_t _pad;
    double alpha;                 /* rel.model.alpha                                         */
    double mean_value;            /* rel.model.mean_value                                    */
    const double *linear_values;  /* dev, nullable: rel.temp.linear_values in COO order      */
    const double *factors[BDF_MAX_MODES]; /* dev: D x N_k sample of every mode of rel; [mode] ignored */
    const double *alpha_dev;      /* dev, nullable: rel.model.alpha in device memory (sampled there, bdf_sample_alpha): read instead of `alpha` */
} bdf_term;

/* sample_latent_all2! (src/sampling.jl:149-172) and sample_user2_all! (:251-264):
 * for every listed row i:  P_i = Lambda + sum_r alpha_r sum_obs w w',  b_i = Lambda mu_i + sum_r alpha_r
 * sum_obs w (y - base),  out[:,i] = chol(inv(P_i))' z + inv(P_i) b_i  with z from stream
 * (BDF_P_ROW, entity_tag, i).  mu: dev, D doubles (shared prior mean) or D x N (mu_is_matrix,
 * macau.jl:103-105).  (shard, n_shards): the rows sampled are positions shard, shard + n_shards, ... of
 * bdf_relation_order(terms[0].rel, terms[0].mode) -- the reference deals rows i:P:N to its P workers
 * (sampling.jl:154); (0, 1) = every row.  out: dev D x N; only this shard's rows are written; must not
 * alias any terms[].factors[k] with k != mode.
 * prior_pack (dev, nullable): the pack bdf_hyper_sample wrote for exactly this (mu, Lambda) -- Lambda mu and Lambda
 * laid out for the row kernel; saves the small pre-launch that otherwise derives them.  Ignored with mu_is_matrix. */
int bdf_sample_rows(bdf_ctx *ctx, int D, int64_t N, int n_terms, const bdf_term *terms,
                    const double *mu, int mu_is_matrix, const double *Lambda,
                    uint32_t entity_tag, int shard, int n_shards, double *out, const double *prior_pack);
/* sample_users_blocked (src/sampling.jl:236-249): the nu users of a Block all observed the same nv items and share ONE
 * covariance inv(Lambda + alpha MM MM'), MM = factor[:, vx]: it is accumulated and factored once for the block, every user
 * then costs its right-hand side (alpha MM Yma[:, u] + Lambda mu) and two triangular solves.  vx_dev: dev nv item ids
 * (0-based); Yma: dev nv x nu column-major (values without their mean, Block.Yma); factor: dev D x M sample of the other side;
 * out: dev D x nu, column u drawn with the normals of stream (BDF_P_ROW, entity_tag, row u).  Same value as the reference's
 * expression (and as bdf_sample_rows on the block as a dense relation) for the same normals. */
int bdf_sample_block(bdf_ctx *ctx, int D, int64_t nu, int64_t nv, const int32_t *vx_dev, const double *Yma,
                     const double *factor, double alpha, const double *mu, const double *Lambda, uint32_t entity_tag,
                     double *out);
/* doubles in a prior pack for num_latent = D */
int bdf_prior_pack_doubles(int D);
/* parity hook: the deterministic part only.  P_out: dev D x D x N, b_out: dev D x N */
int bdf_row_system(bdf_ctx *ctx, int D, int64_t N, int n_terms, const bdf_term *terms,
                   const double *mu, int mu_is_matrix, const double *Lambda,
                   double *P_out, double *b_out);
/* parity hook: number of split rows (rows cut into several work items) that the launches so far left unfinished: 0 */
int bdf_rows_unfinished(bdf_ctx *ctx, int64_t *count);
/* parity hook: n standard normals per row of stream (purpose, entity_tag, row) -> dev n x n_rows */
int bdf_normals(bdf_ctx *ctx, uint32_t purpose, uint32_t entity_tag, int64_t row_begin,
                int64_t n_rows, int n, double *out);
/* parity hook: raw Philox4x32-10 block for (purpose, entity_tag, row, pair) at the current sweep */
int bdf_philox(bdf_ctx *ctx, uint32_t purpose, uint32_t entity_tag, uint64_t row, uint32_t pair,
               uint32_t out_host[4]);

/* ---- a15: hyperprior  (src/sampling.jl:116-127, src/normal_wishart.jl:38-42, macau.jl:120-134) */
/* sumU (dev D) = sum_i U[:,i], UUt (dev D x D) = U U' with U = sample - uhat (uhat nullable);
 * deterministic summation order. */
int bdf_hyper_sums(bdf_ctx *ctx, int D, int64_t N, const double *sample, const double *uhat,
                   double *sumU, double *UUt);
/* Several ranks: the sums over the rows THIS rank owns (chunk c of rank p: positions [(c world + p) cmax, + cmax) of the
 * N = chunks x world x cmax rows, bdf_layout_build) and the ranks' D + D^2 partial sums gathered and added in rank order: the
 * same bits on every rank (src/sampling.jl:117-119 on the master).  comm NULL or one rank: bdf_hyper_sums. */
int bdf_hyper_sums_ranks(bdf_ctx *ctx, bdf_comm *comm, int D, int64_t N, int chunks, const double *sample, const double *uhat,
                         double *sumU, double *UUt);
/* ConditionalNormalWishart + rand(::NormalWishart): draws (mu, Lambda) on the device from the sums.
 * mu0 (dev D), Tinv (dev D x D), b0, nu: the hyper-prior AFTER the feature terms of macau.jl:124-129.
 * params_out (dev, nullable): mu_N (D) followed by inv(T_N) (D x D, the matrix sampling.jl:124 inverts)
 * for parity checks.  prior_pack_out (dev, nullable, bdf_prior_pack_doubles(D) doubles): what bdf_sample_rows needs of
 * the drawn (mu, Lambda), see there.  draws (dev, nullable): output of bdf_hyper_draws for the same arguments. */
int bdf_hyper_sample(bdf_ctx *ctx, int D, int64_t N, const double *sumU, const double *UUt,
                     const double *mu0, double b0, const double *Tinv, double nu,
                     uint32_t entity_tag, double *mu_out, double *Lambda_out, double *params_out, double *prior_pack_out,
                     const double *draws);
/* The random part of bdf_hyper_sample (Bartlett matrix D x D row-major, then the D normals of the mean: D*D + D doubles),
 * which does not depend on the rows: call it for the same (sweep, N, nu, entity_tag) before the rows are done and pass the
 * buffer as `draws` to take the gamma rejection loops off the critical path.  Same streams, same values. */
int bdf_hyper_draws(bdf_ctx *ctx, int D, int64_t N, double nu, uint32_t entity_tag, double *draws_out);

/* Store the pairs sorted by their id in `mode` (stable sort): consecutive pairs then share that mode's factor row, which
 * halves the gather traffic of bdf_predict / bdf_predict_update.  Call before the first update.  The caller's order is
 * kept wherever the interface is per pair: bdf_predict's out and the baseline are indexed through the permutation;
 * bdf_pairs_state returns the running state in STORAGE order, and bdf_pairs_order gives, for every storage position,
 * the caller's index (identity when the pairs were never sorted). */
int bdf_pairs_sort(bdf_pairs *pairs, int mode);
int bdf_pairs_order(const bdf_pairs *pairs, int64_t *orig_host);
/* per-pair baseline (dev, n doubles, borrowed; NULL to clear) that replaces mean_value in bdf_predict / bdf_predict_update for
 * these pairs: mean_value + F_test beta of pred(r, probe_vec, F) (src/sampling.jl:9-14) for a relation with features */
int bdf_pairs_set_baseline(bdf_pairs *pairs, const double *baseline);
/* out (dev, rows of F) = mean_value + F beta, beta dev numF: linear_values (macau.jl:91) / the baseline above */
int bdf_feat_linear(bdf_ctx *ctx, const bdf_feat *F, const double *beta, double mean_value, double *out);
/* sum over the pairs of (value - pred)^2, pred = udot + (linear_values[pair] if non-NULL else mean_value): err' err of
 * sample_alpha (macau.jl:86-87).  stats_out (dev, 4 doubles) as bdf_predict_update's; [1] is the sum.  No running state. */
int bdf_predict_sse(bdf_ctx *ctx, const bdf_pairs *pairs, int D, const double *const *factors, double mean_value,
                    const double *linear_values, double *stats_out);

/* ---- f1/f4: relation model (src/macau.jl:83-92) ---------------------------------------------- */
/* sample_alpha (src/sampling.jl:129-134): alpha ~ Wishart(alpha_nu0 + n, 1 / (1/alpha_lambda0 + sum err^2)) in one dimension.
 * sumsq_err (dev, 1 double): sum over the relation's n observations of (pred - value)^2; alpha_out (dev, 1 double).
 * Gamma stream (BDF_P_GAMMA_N/U, entity 0x800000 | rel_tag, row 0). */
int bdf_sample_alpha(bdf_ctx *ctx, double alpha_lambda0, double alpha_nu0, int64_t n, const double *sumsq_err,
                     uint32_t rel_tag, double *alpha_out);
/* sample_beta_rel (src/sampling.jl:322-337) + linear_values (macau.jl:91): relation-level side information F (one row per
 * observation, COO order), FF path (the reference has no other):
 *   beta = (alpha F'F + lambda_beta I) \ (alpha F'(values - udot - mean_value + alpha^-1/2 z1) + sqrt(lambda_beta) z2)
 *   linear_out = mean_value + F beta
 * train: the relation's observations as pairs (bdf_pairs_create on the COO ids and values); factors as for bdf_predict.
 * beta_out dev numF, linear_out dev nnz, rhs_out dev numF nullable (the right-hand side, for parity checks).
 * z1: stream (BDF_P_BETA_REL1, 0x800000 | rel_tag, row = observation), z2: (BDF_P_BETA_REL2, ..., row = feature). */
int bdf_sample_beta_rel(bdf_ctx *ctx, const bdf_feat *F, const bdf_pairs *train, int D, const double *const *factors,
                        double mean_value, double alpha, double lambda_beta, uint32_t rel_tag,
                        double *beta_out, double *linear_out, double *rhs_out);
/* The same over several ranks (the reference computes err and F'v on the master, macau.jl:83-92): rank r holds a
 * block of consecutive observations [first_obs, first_obs + n) -- F is that block of the feature matrix's rows, train the same
 * observations as pairs.  F'v and, once, F'F are summed over the ranks in rank order (bdf_sum_ranks), every rank solves the
 * numF x numF system and ends with the same beta; z1 is keyed by the observation's place in the whole relation, so the chain
 * does not depend on the number of ranks up to the summation order.  linear_out: this rank's n values (the host gathers the
 * blocks with bdf_allgather_block).  comm NULL or one rank: bdf_sample_beta_rel. */
int bdf_sample_beta_rel_ranks(bdf_ctx *ctx, bdf_comm *comm, const bdf_feat *F, const bdf_pairs *train, int64_t first_obs, int D,
                              const double *const *factors, double mean_value, double alpha, double lambda_beta, uint32_t rel_tag,
                              double *beta_out, double *linear_out, double *rhs_out);
/* x (dev, n doubles) := sum over the ranks of x, added block after block in rank order: every rank ends with the same bits
 * (the squared-error sum of sample_alpha over the ranks' blocks of observations; F'v above).  comm NULL or one rank: no-op. */
int bdf_sum_ranks(bdf_ctx *ctx, bdf_comm *comm, double *x, int64_t n);


/* ---- f2: test-set prediction (src/sampling.jl:9-45, macau.jl:142-203, 231-241) -------- */
/* ids: n x n_modes column-major 1-based (test_vec[:,1:end-1]); values: n (test_vec[:,end]) */
int bdf_pairs_create(bdf_ctx *ctx, int n_modes, int64_t n, const void *ids, int id_bytes,
                     const double *values, bdf_pairs **out);
int bdf_pairs_destroy(bdf_pairs *p);
/* pred(r, test_vec) = udot + mean_value -> out (dev n) */
int bdf_predict(bdf_ctx *ctx, const bdf_pairs *p, int D, const double *const *factors,
                double mean_value, double *out);
/* pred_all(r) (src/sampling.jl:91-97; macau.jl:145-147 accumulates it into predictions_full): udot over EVERY cell of the relation
 * + mean_value.  dims: n_modes (2 .. 4) sizes; factors[k]: dims[k] x D row-major (dev); out (dev): prod(dims) doubles, the cell
 * (i_1, ..., i_n) at ((i_1 dims[1] + i_2) dims[2] + ...) + i_n -- the last mode fastest. */
int bdf_predict_all(bdf_ctx *ctx, int n_modes, const int64_t *dims, int D, const double *const *factors,
                    double mean_value, double *out);
/* one macau.jl:142-203 reporting step: p = pred; phase 0 (burn-in): avg = p; phase 1 (first
 * posterior sample): avg = p, sq = p^2, count = 1; phase 2: running mean / sum of squares.
 * stats_out (dev 4 doubles): sum (y-clamp(avg))^2, sum (y-clamp(p))^2, #correct(avg), #correct(p).
 * clamp_lo > clamp_hi means no clamping. */
int bdf_predict_update(bdf_ctx *ctx, bdf_pairs *p, int D, const double *const *factors,
                       double mean_value, int phase, double clamp_lo, double clamp_hi,
                       double class_cut, double *stats_out);
/* running state: avg (dev n), sq (dev n) */
int bdf_pairs_state(const bdf_pairs *p, double **avg, double **sq, int64_t *n);

/* ---- a8-a14: side information (Entity.F operator contract) ------------- */
/* dense: F host N x numF column-major (RelationData.jl:66-90 `F`) */
int bdf_feat_create_dense(bdf_ctx *ctx, int64_t m, int64_t n, const double *F, bdf_feat **out);
/* SparseMatrixCSR (src/parallel_csr.jl:36-54): COO triplets, 1-based */
int bdf_feat_create_csr(bdf_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int32_t *rows,
                        const int32_t *cols, const double *vals, bdf_feat **out);
/* SparseBinMatrixCSR / SparseBinMatrix (src/sparsebin_csr.jl:22-37, src/parallel_matrix.jl:19-24):
 * implicit 1.0 values; 1-based Int32 rows/cols */
int bdf_feat_create_bin(bdf_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int32_t *rows,
                        const int32_t *cols, bdf_feat **out);
int bdf_feat_destroy(bdf_feat *f);
/* Several GPUs store an entity's rows at internal positions (bdf_layout_build), and the rows of its F with them.
 * row_ids_host (m entries, nullable to clear): the ORIGINAL id of every row of F (negative: a row nobody owns, all zero) --
 * it keys the per-row noise of bdf_sample_beta (E1, src/sampling.jl:298-300), so that beta does not depend on the layout. */
int bdf_feat_set_row_ids(bdf_feat *f, const int32_t *row_ids_host);
int bdf_feat_size(const bdf_feat *f, int64_t *m, int64_t *n, int64_t *nnz);
/* F*B (transpose=0: B n x ncol -> out m x ncol) or At_mul_B(F,B) (transpose=1: B m x ncol -> out
 * n x ncol); B, out dev column-major (RelationData.jl:314-329, parallel_matrix.jl:520-561) */
int bdf_feat_mul(bdf_ctx *ctx, const bdf_feat *f, const double *B, int ncol, double *out, int transpose);
/* AtA_mul_B! for ncol vectors at once: out = (F'F + lambda I) X (src/parallel_cg.jl:7-14) */
int bdf_feat_AtA_mul(bdf_ctx *ctx, const bdf_feat *f, const double *X, int ncol, double lambda, double *out);
/* uhat = (F beta)' : D x N (F_mul_beta, RelationData.jl:314-320; macau.jl:103,112) and, if
 * mu_matrix_out != NULL, mu_matrix = mu .+ uhat (macau.jl:104,113) */
int bdf_uhat(bdf_ctx *ctx, const bdf_feat *f, int D, const double *beta, const double *mu,
             double *uhat_out, double *mu_matrix_out);
/* hyper-prior feature terms (macau.jl:124-129): Tinv_out = WI + beta' beta * lambda_beta */
int bdf_hyper_feature_terms(bdf_ctx *ctx, int D, int64_t numF, const double *beta, const double *WI,
                            const double *lambda_beta_dev, double *Tinv_out);
/* sample_beta + update_beta! (src/sampling.jl:291-312, 361-370): rhs = F'((sample - mu)' + E1) +
 * sqrt(lb) E2; beta = (F'F + lb I) \ rhs by Cholesky of FF (use_ff, solve_full :314-320) or D
 * simultaneous cg_AtA solves (solve_cg2, parallel_matrix.jl:488-507; cg_AtA parallel_cg.jl:63-94)
 * with per-column stopping ||r|| < tol ||b||, maxiter (<=0: numF).  tol NaN => eps()*numF.
 * lambda_beta lives on the device (lambda_beta_dev, 1 double) so that sample_lambda_beta
 * (sampling.jl:136-142; nu, mu hyper-parameters; enabled by sample_lambda) can update it in place.
 * rhs_out (dev numF x D, nullable), iters_out (dev int32 D, nullable). */
int bdf_sample_beta(bdf_ctx *ctx, const bdf_feat *f, int D, const double *sample, const double *mu,
                    const double *Lambda, double *lambda_beta_dev, int use_ff, double tol, int maxiter,
                    int sample_lambda, double lb_nu, double lb_mu, uint32_t entity_tag,
                    double *beta_out, double *rhs_out, int32_t *iters_out);

/* ---- multi-GPU: rows of every entity shared out over the ranks, exchanged after sampling ---------------------------------
 * The reference: sample_latent_all2! deals the rows i:P:N to P workers and ships every factor matrix to every worker in
 * every call (src/sampling.jl:154-171, remotecall_fetch(sample_latent_range_ref, ...)).  Here: one process per GPU; every
 * rank holds a replica of every factor matrix, the observations of ITS rows only, and after sampling its rows of an entity
 * takes part in one in-place all-gather per chunk.
 *
 * bdf_layout_build (host-only): internal row positions of an entity.  Rows in falling order of `degree` (stable; the
 * observations of the row over all the entity's relations) are dealt round-robin to the ranks, a rank's rows round-robin to
 * `chunks` chunks; row i of chunk c of rank p sits at pos = (c * world + p) * cmax + i.  The factor matrix of the entity then
 * has chunks * world * cmax rows (rows nobody owns stay zero), chunk c of it is one contiguous rank-major region.
 * pos_out: N entries; *cmax_out = ceil(ceil(N / world) / chunks). */
int bdf_layout_build(int64_t N, const int64_t *degree, int world, int chunks, int32_t *pos_out, int64_t *cmax_out);
/* bdf_relation_create with a layout per mode (pos[m], cmax[m] from bdf_layout_build with the same world and chunks): the
 * index is the full IndexedDF index as ever; the DEVICE holds only the observations of the rows `rank` owns, addressed by
 * internal positions.  bdf_sample_rows on such a relation takes (shard, n_shards) = (chunk, chunks), N = chunks * world *
 * cmax, writes the rows at their internal positions and keys every row's random stream by its ORIGINAL id, so that the
 * chain does not depend on the number of GPUs (up to the summation order of the hyperprior's sums). */
int bdf_relation_create_sharded(bdf_ctx *ctx, int n_modes, const int64_t *dims, int64_t nnz, const void *ids, int id_bytes,
                                const double *values, const int32_t *const *pos, const int64_t *cmax, int rank, int world,
                                int chunks, bdf_rel **out);
/* The communicator.  RCCL transport: rank 0 calls bdf_comm_unique_id (128 bytes) and hands the id to the other ranks by
 * whatever channel the host has (Julia: its cluster manager; Python: torch.distributed's store); librccl.so is resolved with
 * dlopen at the first call.  Host transport (test rigs with several ranks on one GPU, where RCCL refuses to run): the block is
 * staged through host memory and `fn` -- recv = world blocks of bytes_per_rank, rank-major -- does the exchange. */
#define BDF_COMM_ID_BYTES 128
int bdf_comm_unique_id(void *id_out);
int bdf_comm_create(bdf_ctx *ctx, int rank, int world, const void *unique_id, bdf_comm **out);
typedef int (*bdf_exchange_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int bdf_comm_create_host(bdf_ctx *ctx, int rank, int world, bdf_exchange_fn fn, void *user, bdf_comm **out);
int bdf_comm_destroy(bdf_comm *comm);
int bdf_comm_size(const bdf_comm *comm, int *rank, int *world);
/* LARGE exchanges by DIRECT ALL-PAIRS COPIES over the point-to-point xGMI links instead of RCCL's ring -- the GPU-side answer to
 * the reference shipping the whole sample matrix to every worker (src/sampling.jl:155-171): every rank exports the allocation its
 * block lives in (hipIpcGetMemHandle), opens its peers', and an exchange of bytes_per_rank >= min_bytes is world - 1 concurrent
 * device-to-device copies, one per link, pulled from the owners' mappings (configuration C4 on 8 GPUs: 640 MB per link per users'
 * half-sweep, ~4.2 ms, where a ring moves 7 x 640 MB through one link after the other).  ORDERED ON THE DEVICE: the owner records
 * an interprocess event behind its row kernel, every peer's copy stream waits for it, the caller's stream waits for the copies in
 * bdf_allgather_join -- no stream is synchronised, the next chunk's rows run beside the copies.  `fn` is the host's all-gather
 * (as for bdf_comm_create_host): it carries a control message per rank and exchange (handle, offset, an error code the ranks
 * agree on) and orders the host CALLS (record before wait), nothing else.  Taken by bdf_allgather_rows only (rotating buffers:
 * bdf_comm.hip); added to a communicator of either kind; bdf_comm_disable_peer takes it off again (collectively: every rank or
 * none).  bdf_comm_peer_selftest (collective): one exchange this way whatever its size, complete on return -- what a host runs
 * before it lets the rows take the path.  Exchanges made this way and the bytes this rank pulled: bdf_comm_peer_stats.
 * Unmeasured on several GPUs; tested with two processes on one. */
int bdf_comm_enable_peer(bdf_comm *comm, bdf_exchange_fn fn, void *user, size_t min_bytes);
int bdf_comm_disable_peer(bdf_comm *comm);
int bdf_comm_peer_selftest(bdf_ctx *ctx, bdf_comm *comm, void *buf, size_t bytes_per_rank);
int bdf_comm_peer_stats(const bdf_comm *comm, int64_t *exchanges, int64_t *bytes_pulled);
/* Exchange of chunk `chunk` of the N x D factor matrix `sample` (dev; N = chunks * world * cmax rows, the layout above): an
 * in-place all-gather of the ranks' blocks (ncclAllGather), ordered after the work enqueued so far on ctx's stream, run on
 * the communicator's own stream -- the row kernel of the next chunk runs beside it.  bdf_allgather_join: ctx's stream waits
 * for every exchange enqueued so far. */
int bdf_allgather_rows(bdf_ctx *ctx, bdf_comm *comm, int D, int64_t N, double *sample, int chunk, int chunks);
int bdf_allgather_join(bdf_ctx *ctx, bdf_comm *comm);
/* the same for any buffer of world equal blocks (dev; rank r's block at buf + r * bytes_per_rank) */
int bdf_allgather_block(bdf_ctx *ctx, bdf_comm *comm, void *buf, size_t bytes_per_rank);
/* bdf_sample_beta on several ranks: solve_cg2 shares the D conjugate-gradient solves out over its workers
 * (src/parallel_matrix.jl:488-507); here every rank forms the right-hand side, solves a contiguous block of ceil(D / world)
 * columns and the blocks (with their iteration counts) are all-gathered: beta, lambda_beta and iters_out end up identical on
 * every rank and equal to the single-rank result.  comm NULL or one rank, or use_ff: exactly bdf_sample_beta. */
int bdf_sample_beta_ranks(bdf_ctx *ctx, bdf_comm *comm, const bdf_feat *f, int D, const double *sample, const double *mu,
                          const double *Lambda, double *lambda_beta_dev, int use_ff, double tol, int maxiter,
                          int sample_lambda, double lb_nu, double lb_mu, uint32_t entity_tag,
                          double *beta_out, double *rhs_out, int32_t *iters_out);

/* ---- a2: one Gibbs iteration enqueued from native code (src/macau.jl:80-203; relation-level side information and alpha
 * sampling excepted: those iterations are enqueued step by step through the entry points above) ------------------------
 * rows of every entity (+ exchange) -> hyperpriors -> test-set prediction update, on three streams (rows: ctx's; the other
 * two are created here, chosen so that they really run beside it).  Hand-overs: events on the kernels' own dispatch
 * packets; and -- when ctx came from bdf_ctx_create_rows with CUs set aside, one rank, BDF_NO_POLL unset -- the row kernels
 * poll a per-entity word the hyperprior draw publishes instead of the row stream waiting for the draw's event.
 * The host pays one call per iteration.  `sweep` numbers key the random streams only: they need not increase. */
typedef struct {
    int64_t N;                    /* rows of the factor matrix (the entity's count; chunks * world * cmax with a layout)   */
    int64_t n_real;               /* the entity's count (N of ConditionalNormalWishart, src/sampling.jl:117)             */
    uint32_t tag;                 /* entity tag of the random streams (1-based entity number)                             */
    int32_t n_terms;              /* relations the entity takes part in                                                    */
    struct {
        const bdf_rel *rel;
        int32_t mode;             /* 0-based mode of this entity in rel                                                    */
        int32_t entity_of_mode[BDF_MAX_MODES];   /* which entity (index into the array) every mode of rel is                */
        double alpha, mean_value;
    } terms[BDF_MAX_TERMS];
    double *sample[3];            /* dev, D x N each: the rows rotate through three buffers; [0] holds the current rows      */
    double *mu, *Lambda, *mu0, *WI, *sumU, *UUt, *params /*nullable*/, *prior_pack, *draws;   /* dev, as in bdf_hyper_sample */
    double b0, nu0;
    /* side information of the entity (Entity.F; NULL: none -- a zeroed tail of the struct is "no features").  With it the
     * iteration runs uhat = (F beta)' and the per-row prior means before the entity's rows (macau.jl:103-104), the feature
     * terms in its hyperprior (macau.jl:124-129) and update_beta! after the rows of every entity (macau.jl:138-140) */
    const bdf_feat *feat;
    double *beta;                 /* dev, numF x D                                                                          */
    double *uhat, *mu_matrix;     /* dev, D x N each                                                                        */
    double *Tinv;                 /* dev, D x D: WI + beta' beta lambda_beta                                                */
    double *lambda_beta;          /* dev, 1                                                                                 */
    int32_t *cg_iters;            /* dev, D (nullable)                                                                      */
    int32_t use_ff;               /* (F'F + lambda I) \ rhs directly (numF <= compute_ff_size) or by conjugate gradients    */
    int32_t sample_lambda_beta, full_lambda_u;
    int32_t _pad;
    double tol;                   /* NaN: eps() * numF                                                                      */
    double lb_nu, lb_mu;          /* hyper-parameters of sample_lambda_beta (Entity.nu, Entity.mu)                          */
} bdf_gibbs_entity;
int bdf_gibbs_create(bdf_ctx *rows_ctx, int D, int n_entities, const bdf_gibbs_entity *entities, bdf_gibbs **out);
int bdf_gibbs_destroy(bdf_gibbs *g);
/* the contexts of the hyperprior and prediction streams (owned by g), e.g. to create the test pairs' running state there */
int bdf_gibbs_contexts(bdf_gibbs *g, bdf_ctx **hyper, bdf_ctx **pred);
int bdf_ctx_stream(const bdf_ctx *ctx, void **stream);
/* test pairs updated at the end of every iteration (macau.jl:142-184); entity_of_mode: which entity every mode of the pairs is;
 * the other arguments as bdf_predict_update's */
int bdf_gibbs_set_test(bdf_gibbs *g, bdf_pairs *pairs, const int32_t *entity_of_mode, double mean_value, double clamp_lo,
                       double clamp_hi, double class_cut, double *stats_dev);
/* The relation model inside the native iteration (src/macau.jl:83-92): for every relation registered here bdf_gibbs_sweep runs,
 * BEFORE the entities' rows and on the row stream, sample_alpha (src/sampling.jl:129-134) when alpha_sample is set -- the squared
 * error over `train` (this rank's block of the relation's observations as pairs, with linear_values as their baseline when the
 * relation has features), summed over the ranks, the draw into alpha_dev -- and sample_beta_rel + linear_values
 * (src/sampling.jl:322-337, macau.jl:89-92) when `feat` is set; the row kernels of the relation's entities then read alpha_dev
 * and `linear` (terms are matched to relations by their bdf_rel).  feat_test / test_baseline: the registered test pairs'
 * baseline mean_value + F_test beta is refreshed after the draw (sampling.jl:9-14).  Relations without either need no entry. */
typedef struct {
    const bdf_rel *rel;
    int32_t entity_of_mode[BDF_MAX_MODES];   /* which entity (index into bdf_gibbs_create's array) every mode of rel is          */
    double mean_value;
    double *alpha_dev;            /* dev, 1 double: the current alpha (initialised by the caller)                              */
    int32_t alpha_sample;
    uint32_t rel_tag;             /* 1-based relation number: keys the random streams (0x800000 | rel_tag)                      */
    double alpha_lambda0, alpha_nu0;
    int64_t nnz;                  /* observations of the whole relation (n of sample_alpha)                                     */
    bdf_pairs *train;             /* this rank's block of the observations (COO order) as pairs                                */
    int64_t first_obs, obs_block; /* its first observation; observations per rank block (linear is world x obs_block long)     */
    const bdf_feat *feat;         /* nullable: the block's rows of the relation's feature matrix                               */
    double *beta;                 /* dev, numF                                                                                  */
    double *linear;               /* dev, linear_values of the whole relation                                                   */
    double lambda_beta;
    const bdf_feat *feat_test;    /* nullable: feature rows of the registered test pairs ...                                    */
    double *test_baseline;        /* ... and their baseline (dev, one double per test pair)                                     */
} bdf_gibbs_relation;
int bdf_gibbs_set_relations(bdf_gibbs *g, int n_relations, const bdf_gibbs_relation *rels);
/* several ranks: exchange every entity's rows after sampling them (NULL: none) */
int bdf_gibbs_set_comm(bdf_gibbs *g, bdf_comm *comm);
/* one iteration.  predict_phase: bdf_predict_update's phase (0 burn-in, 1 first posterior sample, 2 later ones), -1: none */
int bdf_gibbs_sweep(bdf_gibbs *g, uint32_t sweep, int predict_phase);      /* predict_phase 3 (set-up): this sample's statistics only, no running state */
/* which of sample[0..2] holds entity's current rows */
int bdf_gibbs_current(const bdf_gibbs *g, int entity, int *buffer);
/* measurement: the row launch of one entity as bdf_gibbs_sweep makes it, and nothing else (no hyperprior update, exchange or
 * prediction update: the chain's state is not kept consistent) */
int bdf_gibbs_rows_only(bdf_gibbs *g, int entity, uint32_t sweep);
/* set-up: brings the device to its working state without advancing the chain.  Full iterations (rows of every entity,
 * hyperprior chains, beta, the prediction kernel on the registered test pairs WITHOUT running state) with iteration numbers
 * no real iteration uses, for about `milliseconds` (with a communicator: milliseconds / 0.1 iterations, the same count on
 * every rank), then the chain's state -- every entity's current sample, (mu, Lambda), sums, prior pack, draws, beta, uhat,
 * lambda_beta, and whether a draw of an earlier iteration exists -- is put back bit for bit.  The buffers rotate meanwhile:
 * ask bdf_gibbs_current afterwards.  (Row launches alone leave a short run of iterations 5 % slower than this does.) */
int bdf_gibbs_warm_device(bdf_gibbs *g, double milliseconds);
/* (set-up) does the entity's hyperprior draw / beta of an earlier iteration exist (bdf_gibbs_sweep takes the next row launch's
 * prior pack from it and waits for it)?  A host that runs iterations whose results it then discards -- the engine's device
 * warm-up: full iterations, then every buffer put back -- puts these two flags back as well. */
int bdf_gibbs_recorded(const bdf_gibbs *g, int entity, int *hyper, int *beta);
int bdf_gibbs_set_recorded(bdf_gibbs *g, int entity, int hyper, int beta);
/* measurement: (start, stop) events ride on the dispatch of entity's next row kernel (bdf_ctx_time_next_rows) */
int bdf_gibbs_time_rows(bdf_gibbs *g, int entity, void *start, void *stop);
int bdf_gibbs_span_rows(bdf_gibbs *g, int entity, void *slot_dev);      /* bdf_ctx_span_next_rows for the next row launch of `entity` */
int bdf_gibbs_sync(bdf_gibbs *g);     /* waits for the three streams; errors as bdf_ctx_sync */

/* ---- synthetic sparse relation of configuration C4 (host-only, needs no GPU) ---------------------------------------
 * The reference's large-scale benchmark draws its relation with sprand (test/benchmark_parallel_latent.jl:8-12).
 * Observations k_begin .. k_end-1 of the relation `seed`: row uniform on 1..n_rows; column Zipf-like, p(c) ~ 1/(c + zipf_offset)
 * (continuous inverse CDF; zipf_offset 0 = uniform); value clip(round(3.5 + <u*_row, v*_col> + 0.5 eps), 1, 5) from a planted
 * rank-8 model with 0.5 N(0,1) factors; held_out[k] = 1 for a test_fraction of the observations (nullable).  Counter-based
 * (Philox, key = seed): observation k does not depend on the range or the number of threads, so every rank of a multi-GPU
 * run can generate the relation (or its part) independently.  rows/cols 1-based int32. */
int bdf_synth_ratings(uint64_t seed, int64_t n_rows, int64_t n_cols, int64_t k_begin, int64_t k_end,
                      double zipf_offset, double test_fraction, int32_t *rows_out, int32_t *cols_out,
                      double *vals_out, uint8_t *held_out);

#ifdef __cplusplus
}
#endif
#endif /* BDF_H */
