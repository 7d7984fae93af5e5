/* bpmf_c_abi.c -- the whole BPMF Gibbs loop of src/macau.jl:80-203 through the C ABI of include/bdf.h alone: no Python,
 * no torch -- what a Julia host does with ccall (julia/BDFHip.jl).  Synthetic ratings from the library's own generator
 * (bdf_synth_ratings, configuration C4's shape at a small size), held-out RMSE printed at the end.
 *
 *   gcc -O2 -std=c11 -Iinclude examples/bpmf_c_abi.c -o bpmf_c_abi -Lbayesiandatafusion.jl_amd/csrc -lbdf_hip -lm \
 *       -Wl,-rpath,$PWD/bayesiandatafusion.jl_amd/csrc
 *   ./bpmf_c_abi [rows cols nnz D sweeps]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bdf.h"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc__ = (call);                                                                \
        if (rc__ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, bdf_last_error()); exit(1); } \
    } while (0)

static void *dev_zeros(bdf_ctx *ctx, size_t doubles)
{
    void *p;
    double *z = (double *)calloc(doubles ? doubles : 1, sizeof(double));
    CHECK(bdf_dev_alloc(ctx, doubles * sizeof(double), &p));
    CHECK(bdf_h2d(ctx, p, z, doubles * sizeof(double)));
    free(z);
    return p;
}

static void *dev_copy(bdf_ctx *ctx, const double *h, size_t doubles)
{
    void *p;
    CHECK(bdf_dev_alloc(ctx, doubles * sizeof(double), &p));
    CHECK(bdf_h2d(ctx, p, h, doubles * sizeof(double)));
    return p;
}

int main(int argc, char **argv)
{
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 20000, n_cols = argc > 2 ? atoll(argv[2]) : 3000;
    const int64_t nnz = argc > 3 ? atoll(argv[3]) : 400000;
    const int D = argc > 4 ? atoi(argv[4]) : 32, sweeps = argc > 5 ? atoi(argv[5]) : 20;
    const double alpha = 2.0;

    /* ---- data: COO triplets, 1 % held out ---- */
    int32_t *rows = malloc(nnz * sizeof(int32_t)), *cols = malloc(nnz * sizeof(int32_t));
    double *vals = malloc(nnz * sizeof(double));
    uint8_t *held = malloc(nnz);
    CHECK(bdf_synth_ratings(777, n_rows, n_cols, 0, nnz, 100.0, 0.01, rows, cols, vals, held));
    int64_t ntrain = 0, ntest = 0;
    for (int64_t k = 0; k < nnz; k++) held[k] ? ntest++ : ntrain++;
    int32_t *ids = malloc(2 * ntrain * sizeof(int32_t)), *tids = malloc(2 * (ntest ? ntest : 1) * sizeof(int32_t));   /* column-major nnz x 2 */
    double *v = malloc(ntrain * sizeof(double)), *tv = malloc((ntest ? ntest : 1) * sizeof(double));
    double mean = 0.0;
    for (int64_t k = 0, a = 0, b = 0; k < nnz; k++) {
        if (held[k]) { tids[b] = rows[k]; tids[ntest + b] = cols[k]; tv[b++] = vals[k]; }
        else { ids[a] = rows[k]; ids[ntrain + a] = cols[k]; v[a++] = vals[k]; mean += vals[k]; }
    }
    mean /= (double)ntrain;

    /* ---- device objects ---- */
    bdf_ctx *ctx;
    CHECK(bdf_ctx_create_rows(0, 42, 8, &ctx));                 /* row stream; 8 CUs left to the hyperprior stream */
    const int64_t dims[2] = {n_rows, n_cols};
    bdf_rel *rel;
    CHECK(bdf_relation_create(ctx, 2, dims, ntrain, ids, 4, v, &rel));
    double rmean;
    CHECK(bdf_relation_value_mean(rel, &rmean));
    bdf_gibbs_entity ent[2];
    memset(ent, 0, sizeof(ent));
    double *eye5 = calloc((size_t)D * D, sizeof(double)), *eye = calloc((size_t)D * D, sizeof(double));
    for (int i = 0; i < D; i++) { eye5[i * D + i] = 5.0; eye[i * D + i] = 1.0; }
    for (int j = 0; j < 2; j++) {
        bdf_gibbs_entity *e = &ent[j];
        e->N = e->n_real = dims[j]; e->tag = (uint32_t)(j + 1); e->n_terms = 1;
        e->terms[0].rel = rel; e->terms[0].mode = j; e->terms[0].entity_of_mode[0] = 0; e->terms[0].entity_of_mode[1] = 1;
        e->terms[0].alpha = alpha; e->terms[0].mean_value = rmean;
        for (int b = 0; b < 3; b++) e->sample[b] = dev_zeros(ctx, (size_t)dims[j] * D);
        e->mu = dev_zeros(ctx, D); e->Lambda = dev_copy(ctx, eye5, (size_t)D * D);          /* EntityModel defaults, RelationData.jl:66-90 */
        e->mu0 = dev_zeros(ctx, D); e->WI = dev_copy(ctx, eye, (size_t)D * D);
        e->sumU = dev_zeros(ctx, D); e->UUt = dev_zeros(ctx, (size_t)D * D); e->params = dev_zeros(ctx, D + (size_t)D * D);
        e->prior_pack = dev_zeros(ctx, bdf_prior_pack_doubles(D)); e->draws = dev_zeros(ctx, (size_t)D * D + D);
        e->b0 = 2.0; e->nu0 = (double)D;
    }
    bdf_gibbs *g;
    CHECK(bdf_gibbs_create(ctx, D, 2, ent, &g));
    bdf_pairs *test = NULL;
    double *stats = dev_zeros(ctx, 4);
    if (ntest) {
        CHECK(bdf_pairs_create(ctx, 2, ntest, tids, 4, tv, &test));
        CHECK(bdf_pairs_sort(test, 1));
        const int32_t eom[2] = {0, 1};
        CHECK(bdf_gibbs_set_test(g, test, eom, rmean, 1.0, 5.0, 2.5, stats));
    }

    /* ---- the Gibbs loop: half burn-in, half posterior samples ---- */
    const int burnin = sweeps / 2;
    for (int i = 1; i <= sweeps; i++)
        CHECK(bdf_gibbs_sweep(g, (uint32_t)i, ntest ? (i <= burnin ? 0 : (i == burnin + 1 ? 1 : 2)) : -1));
    CHECK(bdf_gibbs_sync(g));
    double hs[4] = {0, 0, 0, 0};
    CHECK(bdf_d2h(ctx, hs, stats, sizeof(hs)));
    int cur = 0;
    CHECK(bdf_gibbs_current(g, 0, &cur));
    double *u0 = malloc((size_t)D * sizeof(double));
    CHECK(bdf_d2h(ctx, u0, ent[0].sample[cur], (size_t)D * sizeof(double)));
    double nrm = 0.0;
    for (int d = 0; d < D; d++) nrm += u0[d] * u0[d];
    printf("{\"rows\": %lld, \"cols\": %lld, \"train\": %lld, \"test\": %lld, \"D\": %d, \"sweeps\": %d, \"rmse\": %.6f, "
           "\"accuracy\": %.6f, \"mean\": %.6f, \"row0_norm\": %.12g}\n",
           (long long)n_rows, (long long)n_cols, (long long)ntrain, (long long)ntest, D, sweeps,
           ntest ? sqrt(hs[0] / (double)ntest) : 0.0, ntest ? hs[2] / (double)ntest : 0.0, mean, sqrt(nrm));
    if (test) CHECK(bdf_pairs_destroy(test));
    CHECK(bdf_gibbs_destroy(g));
    CHECK(bdf_relation_destroy(rel));
    CHECK(bdf_ctx_destroy(ctx));
    return 0;
}
