/*
 * bdf_oracle.c -- CPU restatement of the BayesianDataFusion.jl Gibbs-sweep hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libbdf_hip.so) never
 * links, loads or calls anything in this directory.
 *
 * Each function cites the reference file:line (under /root/reference) it restates.
 * The reference is Julia 0.4/0.5; Julia is not installed in the build or GPU image, so
 * the reference itself can be neither run nor compiled here.
 *
 * PARITY STATUS
 *   pinned  : integer index structures (IndexedDF/FastIDF) and the deterministic linear
 *             algebra (SpMV family, AtA_mul_B!, cg_AtA, solve_full, conditional mean /
 *             precision of a row, ConditionalNormalWishart parameters) -- against the
 *             literal known-answer tests the reference holds (tests/test_oracle_*.py cite
 *             test/basic.jl, test/solver.jl, test/sparse_csr.jl, test/sparsebin_csr.jl,
 *             test/sbm.jl, test/parallel_matrix.jl, test/heavy_copyto.jl) and against
 *             independent numpy fp64 algebra.
 *   UNPINNED: the random-number STREAM.  The reference draws from Julia's MersenneTwister
 *             (randn) and Distributions.jl (Wishart, MvNormal, Gamma); none of its tests
 *             stores a seeded value, and neither library can be executed here.  The maps
 *             from standard normals to samples follow the reference exactly
 *             (x = chol(inv(P))' z + inv(P) b, Bartlett Wishart, ...); the normals
 *             themselves come from Philox4x32-10 + Box-Muller.  Sampled quantities are
 *             therefore comparable with the reference by distribution (moments), not by value.
 *
 * Plain C99, no BLAS/LAPACK.  Indices crossing this API are 0-based unless stated.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_D 128

/* ------------------------------------------------------------------------------------ */
/* IndexedDF / FastIDF index (src/IndexedDF.jl:10-21, 46-59)                              */
/* ids: nnz x n_modes, column-major, 1-based (as the DataFrame holds them).              */
/* out: per mode m, rowptr[m][0..dims[m]] and rowids[m][0..nnz-1] holding the 1-based    */
/* COO row numbers in ORIGINAL order, i.e. index[mode][j] = rowids[m][rowptr[j-1]..)     */
/* ------------------------------------------------------------------------------------ */
int orc_index_build(int n_modes, const int64_t *dims, int64_t nnz, const int64_t *ids,
                    int64_t **rowptr, int64_t **rowids)
{
    for (int m = 0; m < n_modes; m++) {
        int64_t N = dims[m];
        int64_t *rp = rowptr[m], *ri = rowids[m];
        memset(rp, 0, sizeof(int64_t) * (size_t)(N + 1));
        const int64_t *col = ids + (size_t)m * (size_t)nnz;
        for (int64_t i = 0; i < nnz; i++) {
            int64_t j = col[i];
            if (j < 1 || j > N) return -1;          /* BoundsError in the reference */
            rp[j]++;
        }
        for (int64_t j = 0; j < N; j++) rp[j + 1] += rp[j];
        int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
        memcpy(fill, rp, sizeof(int64_t) * (size_t)N);
        for (int64_t i = 0; i < nnz; i++) {          /* push!(index[mode][j], i) in row order */
            int64_t j = col[i] - 1;
            ri[fill[j]++] = i + 1;
        }
        free(fill);
    }
    return 0;
}

/* rep_int (src/RelationData.jl:283-291) */
void orc_rep_int(const int64_t *x, const int64_t *times, int64_t n, int64_t *out)
{
    int64_t idx = 0;
    for (int64_t i = 0; i < n; i++)
        for (int64_t t = 0; t < times[i]; t++) out[idx++] = x[i];
}

/* ------------------------------------------------------------------------------------ */
/* Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 v1.09 reference      */
/* constants) + Box-Muller.  Stands in for Julia's randn (src/sampling.jl:211,233,288).  */
/* ------------------------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* stream addressing shared with the HIP library (DESIGN.md "RNG contract") */
void orc_draw(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
              uint64_t row, uint32_t pair, uint32_t out[4])
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t ctr[4] = { (uint32_t)row, (uint32_t)((row >> 32) & 0xffffu) | (pair << 16),
                        sweep, (purpose << 24) | (entity & 0xffffffu) };
    orc_philox4x32_10(ctr, key, out);
}

static double u01(uint32_t lo, uint32_t hi)
{
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * 0x1.0p-53;
}

void orc_uniform_pair(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
                      uint64_t row, uint32_t pair, double u[2])
{
    uint32_t o[4];
    orc_draw(seed, sweep, purpose, entity, row, pair, o);
    u[0] = u01(o[0], o[1]);
    u[1] = u01(o[2], o[3]);
}

void orc_normal_pair(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
                     uint64_t row, uint32_t pair, double z[2])
{
    double u[2];
    orc_uniform_pair(seed, sweep, purpose, entity, row, pair, u);
    double r = sqrt(-2.0 * log(u[0]));
    double t = 6.283185307179586476925286766559 * u[1];
    z[0] = r * cos(t);
    z[1] = r * sin(t);
}

/* n standard normals for (purpose, entity, row): z[2p], z[2p+1] from pair p */
void orc_normals(uint64_t seed, uint32_t sweep, uint32_t purpose, uint32_t entity,
                 uint64_t row, int n, double *z)
{
    for (int p = 0; 2 * p < n; p++) {
        double zz[2];
        orc_normal_pair(seed, sweep, purpose, entity, row, (uint32_t)p, zz);
        z[2 * p] = zz[0];
        if (2 * p + 1 < n) z[2 * p + 1] = zz[1];
    }
}

enum { P_ROW = 1, P_BETA_E1 = 2, P_BETA_E2 = 3, P_NW_NORMAL = 4, P_GAMMA_N = 5, P_GAMMA_U = 6,
       P_NW_MEAN = 7, P_BETA_REL1 = 8, P_BETA_REL2 = 9 };

/* Gamma(shape a, scale 1) -- Marsaglia & Tsang (2000); stands in for Distributions.jl's
 * Gamma/Chisq samplers used by Wishart (src/normal_wishart.jl:39) and sample_lambda_beta
 * (src/sampling.jl:141).  Variate index g addresses the stream; attempt t is the pair. */
double orc_gamma(uint64_t seed, uint32_t sweep, uint32_t entity, uint64_t g, double a)
{
    double boost = 1.0;
    if (a < 1.0) {
        double u[2];
        orc_uniform_pair(seed, sweep, P_GAMMA_U, entity, g, 0xffffu, u);
        boost = pow(u[0], 1.0 / a);
        a += 1.0;
    }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0; t < 256; t++) {
        double z[2], u[2];
        orc_normal_pair(seed, sweep, P_GAMMA_N, entity, g, t, z);
        double x = z[0], v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        orc_uniform_pair(seed, sweep, P_GAMMA_U, entity, g, t, u);
        if (u[0] < 1.0 - 0.0331 * (x * x) * (x * x)) return boost * d * v;
        if (log(u[0]) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
    }
    return boost * d;
}

/* ------------------------------------------------------------------------------------ */
/* small dense helpers (column-major, leading dimension = n)                              */
/* ------------------------------------------------------------------------------------ */
/* general inverse by LU with partial pivoting -- what Julia's inv(::Matrix) does through
 * LAPACK getrf+getri (src/sampling.jl:207,229,284). returns 0 on success */
/* `work`: n * n + n doubles of scratch (NULL: allocated here) */
static int inv_lu_ws(int n, const double *A, double *Ainv, double *work)
{
    double *M = work ? work : (double *)malloc(sizeof(double) * ((size_t)n * n + n));
    double *colk = M + (size_t)n * n;
    memcpy(M, A, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) Ainv[i + (size_t)j * n] = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < n; k++) {
        int p = k; double best = fabs(M[k + (size_t)k * n]);
        for (int i = k + 1; i < n; i++) {
            double v = fabs(M[i + (size_t)k * n]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) { if (!work) free(M); return -1; }
        if (p != k)
            for (int j = 0; j < n; j++) {
                double t = M[k + (size_t)j * n]; M[k + (size_t)j * n] = M[p + (size_t)j * n]; M[p + (size_t)j * n] = t;
                t = Ainv[k + (size_t)j * n]; Ainv[k + (size_t)j * n] = Ainv[p + (size_t)j * n]; Ainv[p + (size_t)j * n] = t;
            }
        double piv = 1.0 / M[k + (size_t)k * n];
        for (int j = 0; j < n; j++) { M[k + (size_t)j * n] *= piv; Ainv[k + (size_t)j * n] *= piv; }
        /* row i -= f_i * row k for every i != k with f_i = M[i][k]: the same products and differences as the row-by-row form,
         * taken column by column so that the inner loop runs along the (column-major) storage */
        for (int i = 0; i < n; i++) colk[i] = (i == k) ? 0.0 : M[i + (size_t)k * n];
        for (int j = 0; j < n; j++) {
            const double mk = M[k + (size_t)j * n], ak = Ainv[k + (size_t)j * n];
            double *Mj = M + (size_t)j * n, *Aj = Ainv + (size_t)j * n;
            for (int i = 0; i < n; i++) {
                if (colk[i] == 0.0) continue;
                Mj[i] -= colk[i] * mk;
                Aj[i] -= colk[i] * ak;
            }
        }
    }
    if (!work) free(M);
    return 0;
}
static int inv_lu(int n, const double *A, double *Ainv) { return inv_lu_ws(n, A, Ainv, NULL); }

/* lower Cholesky factor L (L L' = A) reading the UPPER triangle of A, as
 * chol(Hermitian(covar))' does (src/sampling.jl:211). returns 0 on success */
static int chol_lower_from_upper(int n, const double *A, double *L)
{
    memset(L, 0, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; j++) {
        double s = A[j + (size_t)j * n];
        for (int k = 0; k < j; k++) s -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
        if (!(s > 0.0)) return -1;
        double d = sqrt(s);
        L[j + (size_t)j * n] = d;
        for (int i = j + 1; i < n; i++) {
            double t = A[j + (size_t)i * n];            /* upper-triangle element (j,i) */
            for (int k = 0; k < j; k++) t -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
            L[i + (size_t)j * n] = t / d;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* One relation's contribution to a row (a "term"), host layout.                          */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int n_modes;              /* modes of the relation                                   */
    int mode;                 /* 0-based mode of the entity being sampled                */
    int64_t nnz;
    const int64_t *ids;       /* nnz x n_modes column-major, 1-based (FastIDF.ids)       */
    const double *values;     /* nnz (FastIDF.values)                                    */
    const int64_t *rowptr;    /* index of `mode` (orc_index_build)                       */
    const int64_t *rowids;    /* 1-based COO row numbers                                 */
    double alpha;             /* rel.model.alpha                                         */
    double mean_value;        /* rel.model.mean_value                                    */
    const double *linear_values; /* nullable, per COO row (rel.temp.linear_values)       */
    const double *const *factors; /* [n_modes] D x N_k column-major (entity.model.sample) */
} orc_term;

/* deterministic part of a row's conditional: P = Lambda + sum alpha MM MM',
 * b = Lambda mu + sum alpha MM rr  (src/sampling.jl:266-283; 205-208; 217-230) */
void orc_row_system(int D, int n_terms, const orc_term *terms, int64_t row,
                    const double *mu_i, const double *Lambda, double *P, double *b)
{
    memcpy(P, Lambda, sizeof(double) * (size_t)D * D);
    for (int i = 0; i < D; i++) {
        double s = 0.0;
        for (int j = 0; j < D; j++) s += Lambda[i + (size_t)j * D] * mu_i[j];
        b[i] = s;
    }
    double w[ORC_MAX_D];
    for (int r = 0; r < n_terms; r++) {
        const orc_term *t = &terms[r];
        for (int64_t q = t->rowptr[row]; q < t->rowptr[row + 1]; q++) {
            int64_t o = t->rowids[q] - 1;
            double rr = t->values[o] - (t->linear_values ? t->linear_values[o] : t->mean_value);
            int first = 1;
            for (int k = 0; k < t->n_modes; k++) {
                if (k == t->mode) continue;
                const double *v = t->factors[k] + (size_t)(t->ids[o + (size_t)k * t->nnz] - 1) * D;
                if (first) { for (int d = 0; d < D; d++) w[d] = v[d]; first = 0; }
                else       { for (int d = 0; d < D; d++) w[d] *= v[d]; }   /* MM .*= ... */
            }
            for (int j = 0; j < D; j++) {
                double aw = t->alpha * w[j];
                for (int i = 0; i < D; i++) P[i + (size_t)j * D] += aw * w[i];
                b[j] += aw * rr;
            }
        }
    }
}

/* sample_user_basic / sample_user2 (src/sampling.jl:200-212, 215-234, 266-289), literally:
 * covar = inv(P); mu = covar*b; x = chol(Hermitian(covar))' * z + mu.
 * mean_out (nullable) receives covar*b. returns 0 on success */
/* work: 4 D^2 + D doubles of scratch (NULL: allocated here) */
static int orc_sample_row_ws(int D, int n_terms, const orc_term *terms, int64_t row, const double *mu_i,
                             const double *Lambda, const double *z, double *x, double *mean_out, double *work)
{
    double *P = work ? work : (double *)malloc(sizeof(double) * ((size_t)D * D * 4 + D));
    double *covar = P + (size_t)D * D, *L = covar + (size_t)D * D;
    double b[ORC_MAX_D], m[ORC_MAX_D];
    orc_row_system(D, n_terms, terms, row, mu_i, Lambda, P, b);
    int rc = inv_lu_ws(D, P, covar, L + (size_t)D * D);
    if (!rc) {
        for (int i = 0; i < D; i++) {
            double s = 0.0;
            for (int j = 0; j < D; j++) s += covar[i + (size_t)j * D] * b[j];
            m[i] = s;
        }
        rc = chol_lower_from_upper(D, covar, L);
    }
    if (!rc) {
        for (int i = 0; i < D; i++) {
            double s = m[i];
            for (int j = 0; j <= i; j++) s += L[i + (size_t)j * D] * z[j];
            x[i] = s;
            if (mean_out) mean_out[i] = m[i];
        }
    }
    if (!work) free(P);
    return rc;
}
int orc_sample_row(int D, int n_terms, const orc_term *terms, int64_t row, const double *mu_i,
                   const double *Lambda, const double *z, double *x, double *mean_out)
{
    return orc_sample_row_ws(D, n_terms, terms, row, mu_i, Lambda, z, x, mean_out, NULL);
}

/* sample_latent_range / sample_user2_all! (src/sampling.jl:181-198, 251-264): every row in
 * [row_begin,row_end) with normals from stream (P_ROW, entity_tag, row).  mu is D (shared)
 * or D x N (per row) as in macau.jl:102-107.  `out` is the entity's D x N sample matrix;
 * terms[].factors must NOT alias it for the sampled mode (the reference blanks that slot,
 * sampling.jl:156).  nthreads>1 = the latent_pids data-parallel path (sampling.jl:154). */
int orc_sample_rows(int D, int64_t row_begin, int64_t row_end, int n_terms, const orc_term *terms,
                    const double *mu, int mu_is_matrix, const double *Lambda,
                    uint64_t seed, uint32_t sweep, uint32_t entity_tag, double *out, int nthreads)
{
    int fail = 0;
    /* (one scratch block per thread: a malloc per row serialises 256 threads on the allocator) */
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        double *work = (double *)malloc(sizeof(double) * ((size_t)D * D * 4 + D));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (int64_t i = row_begin; i < row_end; i++) {
            double z[ORC_MAX_D + 1];
            orc_normals(seed, sweep, P_ROW, entity_tag, (uint64_t)i, D, z);
            const double *mu_i = mu_is_matrix ? mu + (size_t)i * D : mu;
            if (orc_sample_row_ws(D, n_terms, terms, i, mu_i, Lambda, z, out + (size_t)i * D, NULL, work)) fail = 1;
        }
        free(work);
    }
    (void)nthreads;
    return fail ? -1 : 0;
}

/* ------------------------------------------------------------------------------------ */
/* The SECOND row sampler ("low-rank"), for rows with few observations.  NOT the reference's  */
/* map from normals to the sample -- the same conditional distribution N(inv(P) b, inv(P))    */
/* (src/sampling.jl:200-212) drawn with D + n normals and an n x n solve instead of a D x D    */
/* inverse and factorisation.  The HIP library's k_rows_lr follows this function; that it     */
/* samples the reference's distribution is proved deterministically in                        */
/* tests/test_oracle_known_answers.py (the map is affine in z: x = m + S z; m == inv(P) b and  */
/* S S' == inv(P) to 1e-10 for every n in 0 .. D/2) and by moments on the device.             */
/*                                                                                            */
/*   Lambda = L L' (lower Cholesky);  per observation o of the row (all terms, in order):     */
/*   wt_o = sqrt(alpha_o) L^-1 w_o,  rt_o = sqrt(alpha_o) (y_o - base_o)                      */
/*   e0 = L' mu_i + z[0..D)                          (the prior draw, in L-coordinates)       */
/*   G  = I_n + Wt' Wt,   tau = G^-1 (rt - Wt' e0 - z[D..D+n))                                */
/*   x  = L^-T (e0 + Wt tau)                                                                  */
/* With z = 0: x = mu + Lambda^-1 W' (I + W Lambda^-1 W')^-1 (r - W mu), the posterior mean in  */
/* its Kalman-gain form; the noise part is the sampler of Bhattacharya, Chakraborty & Mallick  */
/* (Biometrika 2016) for N(0, (I + Phi' Phi)^-1), Phi = Wt'.                                   */
/* z: D + n normals.  returns 0 on success, -1 not positive definite, -2 too many observations */
/* ------------------------------------------------------------------------------------ */
#define ORC_LR_MAX_N 64
static int chol_lower_plain(int n, const double *A, double *L)       /* L L' = A (A symmetric, both triangles valid) */
{
    memset(L, 0, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; j++) {
        double s = A[j + (size_t)j * n];
        for (int k = 0; k < j; k++) s -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
        if (!(s > 0.0)) return -1;
        double d = sqrt(s);
        L[j + (size_t)j * n] = d;
        for (int i = j + 1; i < n; i++) {
            double t = A[i + (size_t)j * n];
            for (int k = 0; k < j; k++) t -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
            L[i + (size_t)j * n] = t / d;
        }
    }
    return 0;
}

int64_t orc_row_count(int n_terms, const orc_term *terms, int64_t row)
{
    int64_t n = 0;
    for (int r = 0; r < n_terms; r++) n += terms[r].rowptr[row + 1] - terms[r].rowptr[row];
    return n;
}

/* Lchol: the D x D lower Cholesky factor of Lambda (column-major), computed once per entity by the caller */
static int orc_sample_row_lowrank_L(int D, int n_terms, const orc_term *terms, int64_t row, const double *mu_i,
                                    const double *Lchol, const double *z, double *x)
{
    const int64_t n64 = orc_row_count(n_terms, terms, row);
    if (n64 > ORC_LR_MAX_N) return -2;
    const int n = (int)n64;
    double Wt[ORC_LR_MAX_N][ORC_MAX_D], rt[ORC_LR_MAX_N], e0[ORC_MAX_D], w[ORC_MAX_D];
    int a = 0;
    for (int r = 0; r < n_terms; r++) {
        const orc_term *t = &terms[r];
        const double sa = sqrt(t->alpha);
        for (int64_t q = t->rowptr[row]; q < t->rowptr[row + 1]; q++, a++) {
            int64_t o = t->rowids[q] - 1;
            rt[a] = sa * (t->values[o] - (t->linear_values ? t->linear_values[o] : t->mean_value));
            int first = 1;
            for (int k = 0; k < t->n_modes; k++) {
                if (k == t->mode) continue;
                const double *v = t->factors[k] + (size_t)(t->ids[o + (size_t)k * t->nnz] - 1) * D;
                if (first) { for (int d = 0; d < D; d++) w[d] = v[d]; first = 0; }
                else       { for (int d = 0; d < D; d++) w[d] *= v[d]; }
            }
            /* wt = sqrt(alpha) L^-1 w: forward substitution */
            for (int i = 0; i < D; i++) {
                double s = w[i];
                for (int k = 0; k < i; k++) s -= Lchol[i + (size_t)k * D] * Wt[a][k];
                Wt[a][i] = s / Lchol[i + (size_t)i * D];
            }
            for (int i = 0; i < D; i++) Wt[a][i] *= sa;
        }
    }
    for (int d = 0; d < D; d++) {                       /* e0 = L' mu + u */
        double s = 0.0;
        for (int i = d; i < D; i++) s += Lchol[i + (size_t)d * D] * mu_i[i];
        e0[d] = s + z[d];
    }
    double G[ORC_LR_MAX_N * ORC_LR_MAX_N], LG[ORC_LR_MAX_N * ORC_LR_MAX_N], tau[ORC_LR_MAX_N];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int d = 0; d < D; d++) s += Wt[i][d] * Wt[j][d];
            G[i + (size_t)j * n] = s;
        }
    if (n > 0 && chol_lower_plain(n, G, LG)) return -1;
    for (int i = 0; i < n; i++) {                       /* rho = rt - Wt' e0 - delta; forward solve */
        double s = rt[i] - z[D + i];
        for (int d = 0; d < D; d++) s -= Wt[i][d] * e0[d];
        for (int k = 0; k < i; k++) s -= LG[i + (size_t)k * n] * tau[k];
        tau[i] = s / LG[i + (size_t)i * n];
    }
    for (int i = n - 1; i >= 0; i--) {                  /* backward solve */
        double s = tau[i];
        for (int k = i + 1; k < n; k++) s -= LG[k + (size_t)i * n] * tau[k];
        tau[i] = s / LG[i + (size_t)i * n];
    }
    double qv[ORC_MAX_D];
    for (int d = 0; d < D; d++) {
        double s = e0[d];
        for (int i = 0; i < n; i++) s += Wt[i][d] * tau[i];
        qv[d] = s;
    }
    for (int i = D - 1; i >= 0; i--) {                  /* x = L^-T q */
        double s = qv[i];
        for (int k = i + 1; k < D; k++) s -= Lchol[k + (size_t)i * D] * x[k];
        x[i] = s / Lchol[i + (size_t)i * D];
    }
    return 0;
}

int orc_sample_row_lowrank(int D, int n_terms, const orc_term *terms, int64_t row, const double *mu_i,
                           const double *Lambda, const double *z, double *x)
{
    double *L = (double *)malloc(sizeof(double) * (size_t)D * D);
    int rc = chol_lower_plain(D, Lambda, L);
    if (!rc) rc = orc_sample_row_lowrank_L(D, n_terms, terms, row, mu_i, L, z, x);
    free(L);
    return rc;
}

/* Which numbers of the row's stream (P_ROW, entity_tag, row) are the low-rank sampler's D + n normals.  The HIP kernel gives
 * every observation of a row one of sixteen lanes and every lane the elements d = j, j + 16, ... of a D-vector; a lane makes
 * one Philox block = one pair of normals at a time, so the assignment follows the lanes: with j = d % 16, k = d / 16
 *     u_d     = number 2 (j + 16 (k / 2)) + k % 2                         (pair j + 16 (k / 2), element k % 2)
 *     delta_a = number 2 (16 ceil(ceil(D / 16) / 2) + a / 2) + a % 2      (pairs behind the u's)
 * z: D + n doubles out; scratch: 2 (D + n) + 4 doubles */
void orc_lowrank_normals(uint64_t seed, uint32_t sweep, uint32_t entity_tag, uint64_t row, int D, int n, double *z, double *scratch)
{
    const int DB = (D + 15) / 16, base = 16 * ((DB + 1) / 2);
    const int total = 2 * (base + (n + 1) / 2);
    orc_normals(seed, sweep, P_ROW, entity_tag, row, total, scratch);
    for (int d = 0; d < D; d++) {
        const int j = d % 16, k = d / 16;
        z[d] = scratch[2 * (j + 16 * (k / 2)) + k % 2];
    }
    for (int a = 0; a < n; a++) z[D + a] = scratch[2 * (base + a / 2) + a % 2];
}

/* every row in [row_begin, row_end) as the HIP library samples them with the low-rank sampler switched on: rows of at most
 * lr_max observations by orc_sample_row_lowrank with the normals orc_lowrank_normals assigns, the others by the reference's
 * map (orc_sample_rows) */
int orc_sample_rows_lowrank(int D, int64_t row_begin, int64_t row_end, int n_terms, const orc_term *terms,
                            const double *mu, int mu_is_matrix, const double *Lambda, int lr_max,
                            uint64_t seed, uint32_t sweep, uint32_t entity_tag, double *out, int nthreads)
{
    int fail = 0;
    double *L = (double *)malloc(sizeof(double) * (size_t)D * D);
    if (chol_lower_plain(D, Lambda, L)) { free(L); return -1; }
    if (lr_max > ORC_LR_MAX_N) lr_max = ORC_LR_MAX_N;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        double *work = (double *)malloc(sizeof(double) * ((size_t)D * D * 4 + D));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (int64_t i = row_begin; i < row_end; i++) {
            double z[ORC_MAX_D + ORC_LR_MAX_N + 2], zz[2 * (ORC_MAX_D + ORC_LR_MAX_N) + 4];
            const double *mu_i = mu_is_matrix ? mu + (size_t)i * D : mu;
            const int64_t n = orc_row_count(n_terms, terms, i);
            if (n <= lr_max) {
                orc_lowrank_normals(seed, sweep, entity_tag, (uint64_t)i, D, (int)n, z, zz);
                if (orc_sample_row_lowrank_L(D, n_terms, terms, i, mu_i, L, z, out + (size_t)i * D)) fail = 1;
            } else {
                orc_normals(seed, sweep, P_ROW, entity_tag, (uint64_t)i, D, z);
                if (orc_sample_row_ws(D, n_terms, terms, i, mu_i, Lambda, z, out + (size_t)i * D, NULL, work)) fail = 1;
            }
        }
        free(work);
    }
    free(L);
    (void)nthreads;
    return fail ? -1 : 0;
}

/* ------------------------------------------------------------------------------------ */
/* Hyperprior: ConditionalNormalWishart (src/sampling.jl:116-127) + rand(::NormalWishart) */
/* (src/normal_wishart.jl:38-42); Wishart by Bartlett decomposition as Distributions.jl   */
/* does (A lower: A_ii = sqrt(chi2(nu - i)), A_ij ~ N(0,1) i>j; Lam = (L_T A)(L_T A)').   */
/* U: D x N (already sample - uhat when features).  Tinv, nu: after the macau.jl:124-129  */
/* feature terms.  Outputs: mu_N, T_N (parameters, for parity checks), mu, Lambda (draw). */
/* ------------------------------------------------------------------------------------ */
int orc_hyper_params(int D, int64_t N, const double *U, const double *mu0, double b0,
                     const double *Tinv, double nu, double *mu_N, double *T_N,
                     double *nu_N, double *beta_N)
{
    double *NS = (double *)calloc((size_t)D * D, sizeof(double));
    double NU[ORC_MAX_D];
    for (int i = 0; i < D; i++) NU[i] = 0.0;
    /* one column of N S per task: every element is still the sum over the rows in their order (the same bits as a serial
     * pass), and a 1.5M-row entity no longer costs a second of one core per draw */
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) if (N * (int64_t)D * D > 4000000)
#endif
    for (int j = 0; j < D; j++) {
        double nu_j = 0.0;
        double *col = NS + (size_t)j * D;
        for (int64_t n = 0; n < N; n++) {
            const double *u = U + (size_t)n * D;
            const double uj = u[j];
            nu_j += uj;
            for (int i = 0; i < D; i++) col[i] += u[i] * uj;
        }
        NU[j] = nu_j;
    }
    *nu_N = nu + (double)N;
    *beta_N = b0 + (double)N;
    for (int i = 0; i < D; i++) mu_N[i] = (b0 * mu0[i] + NU[i]) / (b0 + (double)N);
    for (int j = 0; j < D; j++)
        for (int i = 0; i < D; i++)
            NS[i + (size_t)j * D] += Tinv[i + (size_t)j * D] + b0 * mu0[i] * mu0[j] - (*beta_N) * mu_N[i] * mu_N[j];
    /* Symmetric(...) reads the upper triangle */
    for (int j = 0; j < D; j++)
        for (int i = j + 1; i < D; i++) NS[i + (size_t)j * D] = NS[j + (size_t)i * D];
    int rc = inv_lu(D, NS, T_N);
    free(NS);
    return rc;
}

/* mean_map 0: the reference's map, mu = mu_N + chol(inv(Lambda) / beta_N)' z (normal_wishart.jl:40 through Distributions' MvNormal).
 * mean_map 1 (NOT the reference's function of z; the library's default since round 5): mu = mu_N + Z^-T z / sqrt(beta_N) with
 * Z = L_T A the lower-triangular factor of Lambda = Z Z' that the Wishart draw already holds -- another square root of the same
 * covariance inv(beta_N Lambda), so the same conditional law (tests/test_oracle_known_answers.py proves mean and covariance
 * deterministically); it saves the device a second 32-step factorisation on the iteration's critical chain. */
int orc_hyper_draw2(int D, const double *mu_N, double beta_N, const double *T_N, double nu_N,
                    uint64_t seed, uint32_t sweep, uint32_t entity_tag, int mean_map, double *mu, double *Lambda)
{
    size_t DD = (size_t)D * D;
    double *LT = (double *)malloc(sizeof(double) * DD * 5);
    double *A = LT + DD, *Z = A + DD, *cov = Z + DD, *Lc = cov + DD;
    double *Tsym = (double *)malloc(sizeof(double) * DD);
    /* full(Symmetric(T)) (normal_wishart.jl:27): upper triangle mirrored */
    for (int j = 0; j < D; j++)
        for (int i = 0; i < D; i++) Tsym[i + (size_t)j * D] = (i <= j) ? T_N[i + (size_t)j * D] : T_N[j + (size_t)i * D];
    int rc = chol_lower_from_upper(D, Tsym, LT);
    free(Tsym);
    if (rc) { free(LT); return rc; }
    memset(A, 0, sizeof(double) * DD);
    for (int i = 0; i < D; i++) {
        double zr[ORC_MAX_D + 1];
        orc_normals(seed, sweep, P_NW_NORMAL, entity_tag, (uint64_t)i, D, zr);
        for (int j = 0; j < i; j++) A[i + (size_t)j * D] = zr[j];
        A[i + (size_t)i * D] = sqrt(2.0 * orc_gamma(seed, sweep, entity_tag, (uint64_t)i, 0.5 * (nu_N - (double)i)));
    }
    for (int j = 0; j < D; j++)
        for (int i = 0; i < D; i++) {
            double s = 0.0;
            for (int k = j; k <= i; k++) s += LT[i + (size_t)k * D] * A[k + (size_t)j * D];
            Z[i + (size_t)j * D] = s;
        }
    for (int j = 0; j < D; j++)
        for (int i = 0; i < D; i++) {
            double s = 0.0;
            for (int k = 0; k < D; k++) s += Z[i + (size_t)k * D] * Z[j + (size_t)k * D];
            Lambda[i + (size_t)j * D] = s;
        }
    if (mean_map == 1) {
        /* Z' x = z by backward substitution (Z lower triangular with a positive diagonal), mu = mu_N + x / sqrt(beta_N) */
        double z[ORC_MAX_D + 1], x[ORC_MAX_D + 1];
        orc_normals(seed, sweep, P_NW_MEAN, entity_tag, 0, D, z);
        for (int i = D - 1; i >= 0; i--) {
            double s = z[i];
            for (int k = i + 1; k < D; k++) s -= Z[k + (size_t)i * D] * x[k];
            x[i] = s / Z[i + (size_t)i * D];
        }
        for (int i = 0; i < D; i++) mu[i] = mu_N[i] + x[i] / sqrt(beta_N);
        free(LT);
        return 0;
    }
    /* mu ~ MvNormal(mu_N, inv(Symmetric(Lam)) ./ kappa) */
    rc = inv_lu(D, Lambda, cov);
    if (!rc) {
        for (size_t q = 0; q < DD; q++) cov[q] /= beta_N;
        rc = chol_lower_from_upper(D, cov, Lc);
    }
    if (!rc) {
        double z[ORC_MAX_D + 1];
        orc_normals(seed, sweep, P_NW_MEAN, entity_tag, 0, D, z);
        for (int i = 0; i < D; i++) {
            double s = mu_N[i];
            for (int j = 0; j <= i; j++) s += Lc[i + (size_t)j * D] * z[j];
            mu[i] = s;
        }
    }
    free(LT);
    return rc;
}

int orc_hyper_draw(int D, const double *mu_N, double beta_N, const double *T_N, double nu_N,
                   uint64_t seed, uint32_t sweep, uint32_t entity_tag, double *mu, double *Lambda)
{
    return orc_hyper_draw2(D, mu_N, beta_N, T_N, nu_N, seed, sweep, entity_tag, 0, mu, Lambda);
}

/* sample_lambda_beta (src/sampling.jl:136-142): Gamma(shape nux/2, scale 2 mux/nux) */
double orc_sample_lambda_beta(int D, int64_t numF, const double *beta /*numF x D*/,
                              const double *Lambda, double nu, double mu,
                              uint64_t seed, uint32_t sweep, uint32_t entity_tag)
{
    double nux = nu + (double)numF * (double)D;
    /* trace((beta'beta) Lambda) */
    double tr = 0.0;
    for (int i = 0; i < D; i++)
        for (int j = 0; j < D; j++) {
            double s = 0.0;
            for (int64_t f = 0; f < numF; f++) s += beta[f + (size_t)i * numF] * beta[f + (size_t)j * numF];
            tr += s * Lambda[j + (size_t)i * D];
        }
    double mux = mu * nux / (nu + mu * tr);
    return orc_gamma(seed, sweep, entity_tag, (uint64_t)D, 0.5 * nux) * (2.0 * mux / nux);
}

/* sample_alpha (src/sampling.jl:129-134): Wishart_1(nu0+n, 1/(1/lambda0 + e'e)) */
double orc_sample_alpha(double alpha_lambda0, double alpha_nu0, int64_t n, double sumsq_err,
                        uint64_t seed, uint32_t sweep, uint32_t rel_tag)
{
    double SW = 1.0 / (1.0 / alpha_lambda0 + sumsq_err);
    return SW * 2.0 * orc_gamma(seed, sweep, 0x800000u | rel_tag, 0, 0.5 * (alpha_nu0 + (double)n));
}

/* ------------------------------------------------------------------------------------ */
/* Feature operators (the duck-typed Entity.F contract).                       */
/* kind 0: dense N x numF column-major; 1: CSR real (parallel_csr.jl:36-54);              */
/* 2: binary CSR (sparsebin_csr.jl:49-63); 3: binary COO (parallel_matrix.jl:242-267)     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int kind;
    int64_t m, n, nnz;
    const double *dense;      /* kind 0 */
    const int64_t *rowptr;    /* kind 1,2: m+1, 0-based */
    const int32_t *colind;    /* kind 1,2: 0-based      */
    const double *vals;       /* kind 1                 */
    const int32_t *rows, *cols; /* kind 3: 0-based      */
} orc_feat;

void orc_feat_mul(const orc_feat *F, const double *x, double *y)   /* y = F x */
{
    if (F->kind == 0) {
        for (int64_t i = 0; i < F->m; i++) y[i] = 0.0;
        for (int64_t j = 0; j < F->n; j++) {
            const double *c = F->dense + (size_t)j * F->m; double xj = x[j];
            for (int64_t i = 0; i < F->m; i++) y[i] += c[i] * xj;
        }
    } else if (F->kind == 1 || F->kind == 2) {
        for (int64_t r = 0; r < F->m; r++) {
            double t = 0.0;
            for (int64_t q = F->rowptr[r]; q < F->rowptr[r + 1]; q++)
                t += (F->kind == 1 ? F->vals[q] : 1.0) * x[F->colind[q]];
            y[r] = t;
        }
    } else {
        for (int64_t i = 0; i < F->m; i++) y[i] = 0.0;
        for (int64_t q = 0; q < F->nnz; q++) y[F->rows[q]] += x[F->cols[q]];
    }
}

void orc_feat_tmul(const orc_feat *F, const double *x, double *y)  /* y = F' x */
{
    if (F->kind == 0) {
        for (int64_t j = 0; j < F->n; j++) {
            const double *c = F->dense + (size_t)j * F->m; double t = 0.0;
            for (int64_t i = 0; i < F->m; i++) t += c[i] * x[i];
            y[j] = t;
        }
    } else if (F->kind == 1 || F->kind == 2) {
        for (int64_t j = 0; j < F->n; j++) y[j] = 0.0;
        for (int64_t r = 0; r < F->m; r++)
            for (int64_t q = F->rowptr[r]; q < F->rowptr[r + 1]; q++)
                y[F->colind[q]] += (F->kind == 1 ? F->vals[q] : 1.0) * x[r];
    } else {
        for (int64_t j = 0; j < F->n; j++) y[j] = 0.0;
        for (int64_t q = 0; q < F->nnz; q++) y[F->cols[q]] += x[F->rows[q]];
    }
}

/* AtA_mul_B! (src/parallel_cg.jl:7-14): y = (F'F + lambda I) x ; tmp has F.m entries */
void orc_AtA_mul_B(const orc_feat *F, const double *x, double lambda, double *y, double *tmp)
{
    orc_feat_mul(F, x, tmp);
    orc_feat_tmul(F, tmp, y);
    for (int64_t i = 0; i < F->n; i++) y[i] += lambda * x[i];
}

/* cg_AtA (src/parallel_cg.jl:63-94), literally. returns iterations used */
int orc_cg_AtA(const orc_feat *F, const double *b, double lambda, double tol, int maxiter, double *x)
{
    int64_t n = F->n;
    double *r = (double *)malloc(sizeof(double) * (size_t)(3 * n + F->m));
    double *p = r + n, *z = p + n, *tmp = z + n;
    double nb = 0.0;
    for (int64_t i = 0; i < n; i++) nb += b[i] * b[i];
    tol = tol * sqrt(nb);
    for (int64_t i = 0; i < n; i++) { x[i] = 0.0; r[i] = b[i]; p[i] = b[i]; }
    double bkden = 0.0;
    int iter;
    for (iter = 1; iter <= maxiter; iter++) {
        double bknum = 0.0;
        for (int64_t i = 0; i < n; i++) bknum += r[i] * r[i];
        if (sqrt(bknum) < tol) break;
        if (iter > 1) {
            double bk = bknum / bkden;
            for (int64_t i = 0; i < n; i++) p[i] = bk * p[i] + r[i];
        }
        bkden = bknum;
        orc_AtA_mul_B(F, p, lambda, z, tmp);
        double zp = 0.0;
        for (int64_t i = 0; i < n; i++) zp += z[i] * p[i];
        double ak = bknum / zp;
        for (int64_t i = 0; i < n; i++) { x[i] += ak * p[i]; r[i] -= ak * z[i]; }
    }
    free(r);
    return iter - 1;
}

/* solve_full (src/sampling.jl:314-320): (FF + lambda I) \ rhs ; rhs n x nrhs, LU solve */
int orc_solve_full(int64_t n, const double *FF, const double *rhs, int nrhs, double lambda, double *out)
{
    double *M = (double *)malloc(sizeof(double) * (size_t)n * n);
    memcpy(M, FF, sizeof(double) * (size_t)n * n);
    for (int64_t i = 0; i < n; i++) M[i + (size_t)i * n] += lambda;
    memcpy(out, rhs, sizeof(double) * (size_t)n * nrhs);
    for (int64_t k = 0; k < n; k++) {
        int64_t p = k; double best = fabs(M[k + (size_t)k * n]);
        for (int64_t i = k + 1; i < n; i++) { double v = fabs(M[i + (size_t)k * n]); if (v > best) { best = v; p = i; } }
        if (best == 0.0) { free(M); return -1; }
        if (p != k) {
            for (int64_t j = 0; j < n; j++) { double t = M[k + (size_t)j * n]; M[k + (size_t)j * n] = M[p + (size_t)j * n]; M[p + (size_t)j * n] = t; }
            for (int c = 0; c < nrhs; c++) { double t = out[k + (size_t)c * n]; out[k + (size_t)c * n] = out[p + (size_t)c * n]; out[p + (size_t)c * n] = t; }
        }
        for (int64_t i = k + 1; i < n; i++) {
            double f = M[i + (size_t)k * n] / M[k + (size_t)k * n];
            if (f == 0.0) continue;
            for (int64_t j = k; j < n; j++) M[i + (size_t)j * n] -= f * M[k + (size_t)j * n];
            for (int c = 0; c < nrhs; c++) out[i + (size_t)c * n] -= f * out[k + (size_t)c * n];
        }
    }
    for (int c = 0; c < nrhs; c++)
        for (int64_t i = n - 1; i >= 0; i--) {
            double s = out[i + (size_t)c * n];
            for (int64_t j = i + 1; j < n; j++) s -= M[i + (size_t)j * n] * out[j + (size_t)c * n];
            out[i + (size_t)c * n] = s / M[i + (size_t)i * n];
        }
    free(M);
    return 0;
}

/* noise rows e ~ N(0, Lambda^-1) = chol(inv(Lambda))' z, as rand(MvNormal(0, inv(PDMat(Lambda))), n)
 * does (src/sampling.jl:298-300).  E: n x D column-major (the transposed rand(mv,n)'). */
int orc_noise_rows(int D, int64_t n, const double *Lambda, uint64_t seed, uint32_t sweep,
                   uint32_t purpose, uint32_t entity_tag, double *E)
{
    size_t DD = (size_t)D * D;
    double *cov = (double *)malloc(sizeof(double) * DD * 2), *Lc = cov + DD;
    int rc = inv_lu(D, Lambda, cov);
    if (!rc) rc = chol_lower_from_upper(D, cov, Lc);
    if (!rc)
        for (int64_t i = 0; i < n; i++) {
            double z[ORC_MAX_D + 1];
            orc_normals(seed, sweep, purpose, entity_tag, (uint64_t)i, D, z);
            for (int a = 0; a < D; a++) {
                double s = 0.0;
                for (int j = 0; j <= a; j++) s += Lc[a + (size_t)j * D] * z[j];
                E[i + (size_t)a * n] = s;
            }
        }
    free(cov);
    return rc;
}

/* sample_beta (src/sampling.jl:291-312): rhs = F'((U - mu)' + E1) + sqrt(lb) E2 ; solve by
 * solve_full (use_ff) or D independent cg_AtA (solve_cg2, parallel_matrix.jl:488-507).
 * sample: D x N; beta_out, rhs_out: numF x D column-major; iters_out[D] (CG only). */
int orc_sample_beta(const orc_feat *F, int D, const double *sample, const double *mu,
                    const double *Lambda, double lambda_beta, int use_ff, double tol, int maxiter,
                    uint64_t seed, uint32_t sweep, uint32_t entity_tag,
                    double *beta_out, double *rhs_out, int *iters_out)
{
    int64_t N = F->m, numF = F->n;
    if (isnan(tol)) tol = 2.220446049250313e-16 * (double)numF;    /* eps()*numF, :294-296 */
    double *E1 = (double *)malloc(sizeof(double) * ((size_t)N * D + (size_t)numF * D + (size_t)N));
    double *E2 = E1 + (size_t)N * D, *col = E2 + (size_t)numF * D;
    int rc = orc_noise_rows(D, N, Lambda, seed, sweep, P_BETA_E1, entity_tag, E1);
    if (!rc) rc = orc_noise_rows(D, numF, Lambda, seed, sweep, P_BETA_E2, entity_tag, E2);
    if (rc) { free(E1); return rc; }
    double sl = sqrt(lambda_beta);
    for (int d = 0; d < D; d++) {
        for (int64_t i = 0; i < N; i++) col[i] = sample[d + (size_t)i * D] - mu[d] + E1[i + (size_t)d * N];
        orc_feat_tmul(F, col, rhs_out + (size_t)d * numF);
        for (int64_t f = 0; f < numF; f++) rhs_out[f + (size_t)d * numF] += sl * E2[f + (size_t)d * numF];
    }
    if (use_ff) {
        double *FF = (double *)malloc(sizeof(double) * (size_t)numF * numF);
        double *e = (double *)calloc((size_t)numF, sizeof(double)), *fe = (double *)malloc(sizeof(double) * (size_t)N);
        for (int64_t j = 0; j < numF; j++) {           /* FF = full(At_mul_B(F,F)), RelationData.jl:338 */
            e[j] = 1.0; orc_feat_mul(F, e, fe); orc_feat_tmul(F, fe, FF + (size_t)j * numF); e[j] = 0.0;
        }
        rc = orc_solve_full(numF, FF, rhs_out, D, lambda_beta, beta_out);
        free(FF); free(e); free(fe);
    } else {
        if (maxiter <= 0) maxiter = (int)numF;
        for (int d = 0; d < D; d++) {
            int it = orc_cg_AtA(F, rhs_out + (size_t)d * numF, lambda_beta, tol, maxiter, beta_out + (size_t)d * numF);
            if (iters_out) iters_out[d] = it;
        }
    }
    free(E1);
    return rc;
}

/* sample_beta_rel (src/sampling.jl:322-337): relation-level side information, one feature row per observation.
 *   res   = values - udot - mean_value                       (caller)
 *   aFt_y = alpha F'(res + alpha^-1/2 randn(N)) + sqrt(lambda) randn(numF)
 *   beta  = (alpha FF + lambda I) \ aFt_y                     (FF path only: the reference errors out otherwise)
 * Normals: stream (P_BETA_REL1, 0x800000 | rel_tag, row = observation, 0) and (P_BETA_REL2, ..., row = feature, 0). */
int orc_sample_beta_rel(const orc_feat *F, const double *res, double alpha, double lambda_beta,
                        uint64_t seed, uint32_t sweep, uint32_t rel_tag, double *beta_out, double *rhs_out)
{
    int64_t N = F->m, numF = F->n;
    uint32_t tag = 0x800000u | rel_tag;
    double *v = (double *)malloc(sizeof(double) * (size_t)N);
    double sa = 1.0 / sqrt(alpha), sl = sqrt(lambda_beta);
    for (int64_t i = 0; i < N; i++) {
        double z;
        orc_normals(seed, sweep, P_BETA_REL1, tag, (uint64_t)i, 1, &z);
        v[i] = res[i] + sa * z;
    }
    orc_feat_tmul(F, v, rhs_out);
    for (int64_t f = 0; f < numF; f++) {
        double z;
        orc_normals(seed, sweep, P_BETA_REL2, tag, (uint64_t)f, 1, &z);
        rhs_out[f] = alpha * rhs_out[f] + sl * z;
    }
    double *K = (double *)malloc(sizeof(double) * (size_t)numF * numF);
    double *e = (double *)calloc((size_t)numF, sizeof(double)), *fe = (double *)malloc(sizeof(double) * (size_t)N);
    for (int64_t j = 0; j < numF; j++) {               /* FF = full(F'F), RelationData.jl:351 */
        e[j] = 1.0; orc_feat_mul(F, e, fe); orc_feat_tmul(F, fe, K + (size_t)j * numF); e[j] = 0.0;
    }
    for (size_t q = 0; q < (size_t)numF * numF; q++) K[q] *= alpha;
    int rc = orc_solve_full(numF, K, rhs_out, 1, lambda_beta, beta_out);      /* (alpha FF + lambda I) \ aFt_y */
    free(K); free(e); free(fe); free(v);
    return rc;
}

/* udot for test pairs (src/sampling.jl:30-45, 9-14): pred = sum_k prod_modes sample + mean.
 * ids: n x n_modes column-major 1-based */
void orc_predict(int D, int n_modes, int64_t n, const int64_t *ids, const double *const *factors,
                 double mean_value, double *out)
{
    for (int64_t i = 0; i < n; i++) {
        double s = 0.0;
        for (int d = 0; d < D; d++) {
            double p = 1.0;
            for (int k = 0; k < n_modes; k++) p *= factors[k][(size_t)(ids[i + (size_t)k * n] - 1) * D + d];
            s += p;
        }
        out[i] = s + mean_value;
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
