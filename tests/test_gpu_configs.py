"""BASELINE.json configurations C3, C4 and C5 as GPU tests (C1/C2 are in test_gpu_macau.py).

Per configuration: (a) whole Gibbs iterations against the CPU oracle at a reduced size with the configuration's
structure (same number of modes / relations / feature kind / num_latent); (b) at the configuration's full size (C4: a
C4-shaped relation that fits the test budget, with the 64-bit gather the full size needs forced on) properties that
need no oracle -- the union of two shards is bit-equal to the unsharded launch, another item size gives the same rows to
rounding, sampled rows equal the reference's map chol(inv(P_i))' z + inv(P_i) b_i (src/sampling.jl:200-212, 266-289)
recomputed in numpy from the row-system hook, the beta update solves the reference's system (src/sampling.jl:291-320);
(c) the planted / published model quality.
"""
import ctypes as C

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _oracle_feat(O, F):
    """the oracle's operator for an Entity.F (dense array or scipy sparse with unit entries)"""
    if hasattr(F, "tocoo"):
        coo = F.tocsr().tocoo()
        return O.Feat.from_bincsr(coo.row, coo.col, F.shape[0], F.shape[1]), np.asarray(F.todense()) if F.shape[0] * F.shape[1] < 5e7 else None
    F = np.asarray(F, dtype=np.float64)
    return O.Feat.from_dense(F), F


def oracle_macau(O, rd, D, seed, iters, use_ff, lowrank=None):
    """macau.jl:80-140 on the CPU oracle for any RelationData without relation-level features: every entity's rows (sum
    over its relations, tensor relations by Hadamard products), hyperpriors, then the beta updates.  lowrank: {entity index:
    longest row taken by the low-rank sampler} -- the entities whose short rows the library samples that way"""
    ents, rels = rd.entities, rd.relations
    N = [e.count for e in ents]
    S = [np.zeros((n, D)) for n in N]
    mu = [np.zeros(D) for _ in ents]
    Lam = [5.0 * np.eye(D) for _ in ents]
    feats = [None if e.F is None else _oracle_feat(O, e.F) for e in ents]
    beta = [None if f is None else np.zeros((f[0].n, D)) for f in feats]
    lb = [1.0 for _ in ents]
    eidx = lambda e: [x is e for x in ents].index(True)
    index = [O.index_build(r.data.ids, list(r.data.dims)) for r in rels]
    means = [r.data.valueMean() for r in rels]
    for it in range(1, iters + 1):
        for j, en in enumerate(ents):
            terms = []
            for r in en.relations:
                ri = [x is r for x in rels].index(True)
                mode = [e is en for e in r.entities].index(True)
                facs = [None if k == mode else S[eidx(e)] for k, e in enumerate(r.entities)]
                terms.append(O.Term(r.data.ids, r.data.values, list(r.data.dims), mode, r.model.alpha, means[ri], facs, index=index[ri]))
            if feats[j] is not None:
                uhat = np.stack([feats[j][0].mul(beta[j][:, d]) for d in range(D)], axis=1)
                if lowrank and j in lowrank:
                    S[j] = O.sample_rows_lowrank(D, N[j], terms, mu[j] + uhat, Lam[j], lowrank[j], seed, it, j + 1)
                else:
                    S[j] = O.sample_rows(D, N[j], terms, mu[j] + uhat, Lam[j], seed, it, j + 1)
                U, nu, Tinv = S[j] - uhat, D + feats[j][0].n, np.eye(D) + beta[j].T @ beta[j] * lb[j]
            else:
                if lowrank and j in lowrank:
                    S[j] = O.sample_rows_lowrank(D, N[j], terms, mu[j], Lam[j], lowrank[j], seed, it, j + 1)
                else:
                    S[j] = O.sample_rows(D, N[j], terms, mu[j], Lam[j], seed, it, j + 1)
                U, nu, Tinv = S[j], float(D), np.eye(D)
            mu_N, beta_N, T_N, nu_N = O.hyper_params(U, np.zeros(D), 2.0, Tinv, nu)
            mu[j], Lam[j] = O.hyper_draw(mu_N, beta_N, T_N, nu_N, seed, it, j + 1)
        for j in range(len(ents)):
            if feats[j] is not None:
                beta[j], _, _ = O.sample_beta(feats[j][0], S[j], mu[j], Lam[j], lb[j], use_ff, None, seed, it, j + 1)
                lb[j] = O.sample_lambda_beta(beta[j], Lam[j], 1e-3, 1.0, seed, it, j + 1)
    return S, mu, Lam, beta, lb


def _compare(rd, S, mu, Lam, beta, lb, tol=1e-6):
    for j, en in enumerate(rd.entities):
        np.testing.assert_allclose(en.model.sample.T, S[j], rtol=tol, atol=tol, err_msg=f"sample of {en.name}")
        np.testing.assert_allclose(en.model.mu, mu[j], rtol=tol, atol=tol)
        np.testing.assert_allclose(en.model.Lambda, Lam[j], rtol=tol, atol=tol)
        if beta[j] is not None:
            np.testing.assert_allclose(en.model.beta, beta[j], rtol=10 * tol, atol=tol, err_msg=f"beta of {en.name}")
            assert abs(en.lambda_beta - lb[j]) <= 1e-5 * lb[j]


def _shards_items_map(eng, j, D, rows_checked, mu_is_matrix=False, tol=1e-8):
    """full-size properties of the row kernel on entity j of a warmed-up engine: (i) two shards == one launch, bit for bit;
    (ii) item size 64 == default to rounding; (iii) the reference's map recomputed from bdf_row_system + the row's normals"""
    from bdf_amd._lib import check, lib
    st, ctx, terms = eng.ent[j], eng.ctx, eng._terms(j)
    nt = len(terms)
    mu, ism = (st.mu_matrix, 1) if mu_is_matrix else (st.mu, 0)
    ctx.set_sweep(9)

    def rows(shards=1):
        out = ctx.zeros(st.N, D)
        for s in range(shards):
            check(lib().bdf_sample_rows(ctx.handle, D, st.N, nt, terms, _p(mu), ism, _p(st.Lambda), st.tag, s, shards, _p(out), None))
        ctx.sync()
        return out.cpu().numpy()

    auto = rows(1)                   # (the automatic item size: at 16 < D <= 32 a single two-mode relation takes K1c, k_rows_col.hip)
    assert np.all(np.isfinite(auto))
    assert np.array_equal(auto, rows(3)), "union of three shards differs from the unsharded launch (default dispatch)"
    ctx.set_item_size(192)           # (explicit: the comparisons below are between launches of known item sizes, all through k_rows)
    a = rows(1)
    np.testing.assert_allclose(auto, a, rtol=tol, atol=tol)
    assert np.array_equal(a, rows(2)), "union of two shards differs from the unsharded launch"
    ctx.set_item_size(64)
    b = rows(1)
    ctx.set_item_size(192)
    np.testing.assert_allclose(b, a, rtol=tol, atol=tol)
    P_t, b_t, z_t = ctx.zeros(st.N, D, D), ctx.zeros(st.N, D), ctx.zeros(st.N, D)
    check(lib().bdf_row_system(ctx.handle, D, st.N, nt, terms, _p(mu), ism, _p(st.Lambda), _p(P_t), _p(b_t)))
    check(lib().bdf_normals(ctx.handle, 1, st.tag, 0, st.N, D, _p(z_t)))
    ctx.sync()
    import torch
    sel = torch.as_tensor(np.asarray(rows_checked), device=P_t.device)
    P, bb, z = P_t[sel].cpu().numpy(), b_t[sel].cpu().numpy(), z_t[sel].cpu().numpy()
    for q, row in enumerate(rows_checked):
        cov = np.linalg.inv(P[q])
        np.testing.assert_allclose(a[row], np.linalg.cholesky(cov) @ z[q] + cov @ bb[q], rtol=1e-7, atol=1e-8)
    assert ctx.rows_unfinished() == 0
    return a


# ---- C2: BPMF MovieLens-1M, D = 32 (BASELINE.json configs[1], the headline) at FULL size against the oracle chain ------
def test_c2_full_size_whole_iterations_match_oracle(B, O):
    """The bench's workload itself -- MovieLens-1M, 500,209 training ratings, D = 32, default dispatch (K1c for both entities,
    hyperprior chains, the native iteration) -- two whole iterations, every one of the 9,992 sampled rows and both (mu, Lambda)
    against the oracle's chain (src/macau.jl:80-140, sampling.jl:116-127, 200-212) at 1e-6."""
    from bdf_amd import datasets
    from bdf_amd.engine import GibbsEngine
    rd, source = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
    D = 32
    eng = GibbsEngine(rd, D, seed=11)
    assert eng.native
    for i in (1, 2):
        eng.sweep(i)
    eng.sync()
    assert eng.ctx.rows_unfinished() == 0
    _compare(rd, *oracle_macau(O, rd, D, 11, 2, True), tol=1e-6)
    eng.close()


# ---- C3: Macau MovieLens + dense user side information 6040 x 500, D = 32 ---------------------------------------------
@pytest.mark.parametrize("use_ff", [True, False])
def test_c3_full_size_one_iteration_matches_oracle(B, O, use_ff):
    """C3 at its full size (MovieLens + dense F 6040 x 500 i.i.d., D = 32): one whole native iteration -- uhat, rows with per-row
    prior means, hyperprior with the feature terms, the beta update by the direct solve / by CG (parallel_cg.jl:63-94), lambda_beta --
    against the oracle.  beta at 1e-5 (CG columns stop at the reference's own tolerance, eps * numF)."""
    from bdf_amd import datasets
    from bdf_amd.engine import GibbsEngine
    rd, source = datasets.c3_relation_data(B, "iid")
    D = 32
    eng = GibbsEngine(rd, D, seed=7, compute_ff_size=6500 if use_ff else 0)
    assert eng.native and rd.entities[0].use_FF is use_ff
    eng.sweep(1)
    eng.sync()
    eng.sync_host_scalars()          # (lambda_beta back to the Entity: macau() does this at its end)
    _compare(rd, *oracle_macau(O, rd, D, 7, 1, use_ff), tol=1e-6)
    eng.close()


@pytest.mark.parametrize("use_ff", [True, False])
def test_c3_reduced_whole_iterations_match_oracle(B, O, use_ff):
    """C3's structure at a size the oracle finishes in seconds: dense features with numF = 96 > 64 (the FF path's blocked
    direct solve, sampling.jl:314-320, or CG, parallel_cg.jl:63-94), D = 32, lambda_beta sampled"""
    rng = np.random.default_rng(33)
    N1, N2, D, nnz, numF = 400, 300, 32, 14000, 96
    ids = np.stack([rng.integers(1, N1 + 1, nnz), rng.integers(1, N2 + 1, nnz)], axis=1)
    vals = np.clip(np.round(3.5 + rng.standard_normal(nnz)), 1, 5)
    F = rng.standard_normal((N1, numF))
    rel = B.Relation({"u": ids[:, 0], "v": ids[:, 1], "y": vals}, "r", [B.Entity("u", F=F), B.Entity("v")], dims=[N1, N2])
    B.setPrecision(rel, 1.5)
    rd = B.RelationData(rel)
    B.macau(rd, burnin=2, psamples=0, num_latent=D, verbose=False, seed=21, compute_ff_size=6500 if use_ff else 0)
    assert rd._engine.native          # entity side information runs inside the native iteration (bdf_gibbs_sweep)
    _compare(rd, *oracle_macau(O, rd, D, 21, 2, use_ff))


@pytest.mark.parametrize("kind,use_ff", [("iid", True), ("iid", False), ("correlated", True), ("correlated", False)])
def test_c3_full_size_properties(B, kind, use_ff):
    from bdf_amd import datasets
    from bdf_amd._lib import check, lib
    import torch
    rd, source = datasets.c3_relation_data(B, kind)
    D = 32
    eng = B.GibbsEngine(rd, D, seed=7, compute_ff_size=6500 if use_ff else 0)
    for i in range(1, 4):
        eng.sweep(i)
    eng.sync()
    st, en = eng.ent[0], rd.entities[0]
    assert st.numF == 500 and en.use_FF is use_ff
    # the beta update solves (F'F + lambda_beta I) beta = rhs of sampling.jl:298-312: residual from the operator itself
    eng.ctx.set_sweep(5)
    lam0 = float(st.lambda_beta.item())
    beta, rhs, its = eng.ctx.zeros(D, st.numF), eng.ctx.zeros(D, st.numF), torch.zeros(D, dtype=torch.int32, device=eng.ctx.device)
    lb = eng.ctx.tensor([lam0])
    check(lib().bdf_sample_beta(eng.ctx.handle, st.F.handle, D, _p(st.sample), _p(st.mu), _p(st.Lambda), _p(lb), int(use_ff),
                                float("nan"), 0, 0, 1e-3, 1.0, st.tag, _p(beta), _p(rhs), _p(its)))
    Ab = st.F.AtA_mul(beta, lam0)
    eng.ctx.sync()
    res = (Ab - rhs).norm(dim=1) / rhs.norm(dim=1)
    its = its.cpu().numpy()
    if use_ff:
        assert float(res.max()) < 1e-10, float(res.max())          # direct solve (numF <= 640: through F'F = Q diag(s) Q')
        assert its.max() == 0
        # the same solve by the blocked Cholesky factorisation (what larger feature sets take; BDF_NO_EIG forces it here)
        beta2 = eng.ctx.zeros(D, st.numF)
        os.environ["BDF_NO_EIG"] = "1"
        try:
            check(lib().bdf_sample_beta(eng.ctx.handle, st.F.handle, D, _p(st.sample), _p(st.mu), _p(st.Lambda), _p(lb), 1,
                                        float("nan"), 0, 0, 1e-3, 1.0, st.tag, _p(beta2), _p(rhs), _p(its_t := torch.zeros(D, dtype=torch.int32, device=eng.ctx.device))))
        finally:
            del os.environ["BDF_NO_EIG"]
        Ab2 = st.F.AtA_mul(beta2, lam0)
        eng.ctx.sync()
        assert float(((Ab2 - rhs).norm(dim=1) / rhs.norm(dim=1)).max()) < 1e-10
        assert float((beta2 - beta).abs().max()) <= 1e-8 * max(1.0, float(beta.abs().max()))
    else:
        # cg_AtA stops when ||r|| < eps * numF * ||b|| or after numF iterations (parallel_cg.jl:65-75)
        assert float(res.max()) < (1e-9 if kind == "iid" else 1e-6), float(res.max())
        assert 5 <= its.min() and its.max() <= 500
        if kind == "correlated":
            assert its.max() > 30          # 20 dominant directions on top of a flat spectrum: slower than the i.i.d. case
    _shards_items_map(eng, 0, D, (0, 17, 1000, 6039), mu_is_matrix=True)
    eng.close()


def test_c3_quality(B):
    """Macau on MovieLens with 500 uninformative dense user features, D = 32, CG forced (BASELINE configs[2]): the sampled
    lambda_beta shrinks the link matrix, the held-out RMSE stays at BPMF's (0.860 after 20+20)"""
    from bdf_amd import datasets
    rd, source = datasets.c3_relation_data(B, "iid")
    res = B.macau(rd, burnin=20, psamples=20, num_latent=32, verbose=False, clamp=[1.0, 5.0], seed=3, compute_ff_size=0)
    assert rd.entities[0].use_FF is False
    assert 0.84 < res["RMSE"] < 0.885, res["RMSE"]
    assert rd.entities[0].lambda_beta > 1.0


# ---- C5: 3-mode tensor + matrix sharing an entity, binary sparse features, D = 32 --------------------------------------
@pytest.mark.parametrize("use_ff", [False, True])
def test_c5_reduced_whole_iterations_match_oracle(B, O, use_ff):
    from bdf_amd import datasets
    rd, _ = datasets.c5_relation_data(B, nA=600, nB=12, nC=40, nT=30, n1=30000, n2=6000, n_feat=150, feat_per_row=6)
    D = 32
    B.macau(rd, burnin=2, psamples=0, num_latent=D, verbose=False, seed=31, compute_ff_size=6500 if use_ff else 0)
    assert len(rd.entities) == 4 and len(rd.entities[0].relations) == 2 and rd.entities[0].use_FF is use_ff
    assert rd._engine.native
    _compare(rd, *oracle_macau(O, rd, D, 31, 2, use_ff))


def test_c5_full_size_properties_and_quality(B):
    from bdf_amd import datasets
    rd, info = datasets.c5_relation_data(B)
    D = 32
    res = B.macau(rd, burnin=12, psamples=12, num_latent=D, verbose=False, compute_ff_size=0, seed=3)
    # planted rank-8 CP model, noise 0.1 on values of unit scale: the held-out cells are predicted far below the value spread
    assert res["RMSE"] < 0.45 * info["value_std"], (res["RMSE"], info)
    eng = rd._engine
    its = eng.ent[0].cg_iters.cpu().numpy()
    assert 3 <= its.min() and its.max() <= 200
    # entity A: two relations (3-mode + 2-mode), per-row prior means from the binary features, split rows
    _shards_items_map(eng, 0, D, (0, 5, 40000, 99999), mu_is_matrix=True)
    # entity B: 64 rows of ~78,000 observations each (every row split into the maximum number of pieces)
    _shards_items_map(eng, 1, D, (0, 63))
    eng.close()


# ---- C4: large two-mode relation, D = 64 -------------------------------------------------------------------------------
def test_c4_reduced_whole_iterations_match_oracle(B, O):
    """D = 64 with the 64-bit gather offsets the full configuration needs (10M x 64 x 8 B = 5.1 GB factor)"""
    from bdf_amd import datasets
    from bdf_amd.engine import GibbsEngine
    rd = datasets.c4_relation_data(B, 500, 120, 9000, test_fraction=0.0)
    D = 64
    eng = GibbsEngine(rd, D, seed=13)
    eng.ctx.set_gather(2)
    for i in range(1, 3):
        eng.sweep(i)
    eng.sync()
    _compare(rd, *oracle_macau(O, rd, D, 13, 2, True), tol=1e-6)
    eng.close()


def test_c4_dispatch_whole_iterations_match_oracle(B, O):
    """C4's DEFAULT dispatch end to end at a size the oracle sweeps in seconds: 12,000 users of ~10 observations (more than the 8,192
    rows from which the low-rank sampler is taken, and more than half the items' count) x 300 items, D = 64, 64-bit gather offsets
    forced as the full size needs: the users' rows of at most 16 observations through K1-lr (k_rows_lr4<64> + k_rowmat<64> +
    k_lr_prep<64>), those of 17 .. 32 through k_rows_lr32<64> (two observations per lane), the longer ones and the items through
    k_rows<64>, inside the native iteration -- two whole iterations against the oracle dispatching the same way
    (lowrank = {users: 32})."""
    from bdf_amd import datasets
    from bdf_amd.engine import GibbsEngine
    rd = datasets.c4_relation_data(B, 12_000, 300, 120_000, test_fraction=0.0)
    D = 64
    cnt = np.bincount(rd.relations[0].data.ids[:, 0] - 1, minlength=12_000)
    assert (cnt <= 16).sum() >= 10_000 and (cnt > 16).sum() >= 100 and (cnt > 32).sum() == 0
    eng = GibbsEngine(rd, D, seed=13)
    eng.ctx.set_gather(2)
    for i in range(1, 3):
        eng.sweep(i)
    eng.sync()
    assert eng.ctx.rows_unfinished() == 0
    assert eng.rows_dispatch(0)["lowrank"] == 12_000          # every user: at most 16 by k_rows_lr4, 17 .. 32 by k_rows_lr32
    _compare(rd, *oracle_macau(O, rd, D, 13, 2, True, lowrank={0: 32}), tol=1e-6)
    eng.close()


def test_c4_full_size_properties(B):
    """C4 at its REAL size -- 10M x 1M, 100M observations (1 % held out), D = 64, the default dispatch (K1-lr for 9.7M users,
    k_rows<64> with 64-bit offsets for the rest and the items): after two sweeps every sampled row is finite, no split row is left
    unfinished, and for both entities the union of two shards equals the unsharded launch bit for bit.  ~16 GiB of device memory
    plus two 5 GB outputs: skipped on a device with less than 40 GiB free."""
    import torch
    from bdf_amd import datasets
    from bdf_amd._lib import check, lib
    from bdf_amd.engine import GibbsEngine
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2 ** 30:
        pytest.skip("less than 40 GiB of device memory free")
    rd = datasets.c4_relation_data(B)
    D = 64
    eng = GibbsEngine(rd, D, seed=5)
    for i in (1, 2):
        eng.sweep(i)
    eng.sync()
    ctx = eng.ctx
    assert ctx.rows_unfinished() == 0
    ctx.set_sweep(9)
    for j in (1, 0):
        st, terms = eng.ent[j], eng._terms(j)
        assert bool(torch.isfinite(st.sample).all())
        full, halves = ctx.zeros(st.N, D), ctx.zeros(st.N, D)
        check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, _p(st.mu), 0, _p(st.Lambda), st.tag, 0, 1, _p(full), None))
        for sh in range(2):
            check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, _p(st.mu), 0, _p(st.Lambda), st.tag, sh, 2, _p(halves), None))
        ctx.sync()
        assert bool(torch.isfinite(full).all())
        assert torch.equal(full, halves), f"entity {j}: union of two shards differs from the unsharded launch"
        del full, halves
    assert ctx.rows_unfinished() == 0
    eng.close()


def test_c4_shaped_full_size_properties_and_quality(B):
    """a C4-shaped relation that fits the test budget: 1M x 100k, 20M observations (1 % held out), D = 64, 64-bit gather forced"""
    from bdf_amd import datasets
    from bdf_amd._lib import check, lib
    from bdf_amd.engine import GibbsEngine
    rd = datasets.c4_relation_data(B, 1_000_000, 100_000, 20_000_000)
    rel = rd.relations[0]
    D = 64
    eng = GibbsEngine(rd, D, seed=5)
    eng.ctx.set_gather(2)
    test = eng.test_pairs()
    for i in range(1, 61):
        stats = eng.step(i, 0 if i <= 30 else (1 if i == 31 else 2), [1.0, 5.0], rel.class_cut)
    eng.sync()
    rmse = float(np.sqrt(stats.cpu().numpy()[0] / test.n))
    tv = np.asarray(rel.test_vec.values)
    # 30 + 30 sweeps on ~20 ratings per user at D = 64: 0.72 (0.745 after 10 + 10, 0.714 after 60 + 60; tools/c4_quality_probe.py)
    # against 0.88 for the mean predictor; the generator's floor is sqrt(0.25 + 1/12) = 0.58
    assert 0.55 < rmse < 0.75 and tv.std() > 0.85, (rmse, tv.std())
    # items (100k rows, Zipf-like: the head rows are split into the maximum number of pieces), wide gather
    a = _shards_items_map(eng, 1, D, (0, 1, 99, 5000, 99999))
    # the same launch with 32-bit offsets (the factor matrices of this size allow both): identical arithmetic
    st, terms, ctx = eng.ent[1], eng._terms(1), eng.ctx
    ctx.set_gather(0)
    out = ctx.zeros(st.N, D)
    check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, _p(st.mu), 0, _p(st.Lambda), st.tag, 0, 1, _p(out), None))
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), a)
    # users: shards and item sizes only (the row-system dump of 1M rows would be 33 GB)
    ctx.set_gather(2)
    st, terms = eng.ent[0], eng._terms(0)
    full, halves = ctx.zeros(st.N, D), ctx.zeros(st.N, D)
    check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, _p(st.mu), 0, _p(st.Lambda), st.tag, 0, 1, _p(full), None))
    for s in range(3):
        check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, _p(st.mu), 0, _p(st.Lambda), st.tag, s, 3, _p(halves), None))
    ctx.sync()
    import torch
    assert torch.equal(full, halves)
    assert ctx.rows_unfinished() == 0
    eng.close()


@pytest.mark.parametrize("D", [10, 30])
def test_mref_shaped_whole_iterations_match_oracle(B, O, D):
    """The reference's own benchmark shape (test/benchmark_parallel_latent.jl:8-61: sprand(1_500_000, 1000, 0.01), U(0,1)
    values, BPMF D = 10 and 30) at a size the oracle sweeps in seconds: 60,000 x 1,000 with 600,000 observations, two whole
    iterations against the CPU oracle.  At D = 10 the 60,000-row entity is above the row count from which the short rows go four
    to a wave (k_rows_small, default-on), and the 1,000 rows of ~600 observations are split rows of k_rows; at D = 30 the rows
    of ~10 observations take the low-rank sampler (k_rows_lr)."""
    from bdf_amd.engine import GibbsEngine
    rng = np.random.default_rng(1500)
    N, M = 60_000, 1000
    nnz = 600_000
    key = np.unique(rng.integers(0, N * M, size=int(nnz * 1.01)))[:nnz]
    ids = np.stack([key // M + 1, key % M + 1], axis=1)
    vals = rng.random(len(key))
    rel = B.Relation((ids, vals), "r", [B.Entity("rows"), B.Entity("cols")], dims=[N, M])
    rd = B.RelationData(rel)
    eng = GibbsEngine(rd, D, seed=21)
    for i in (1, 2):
        eng.sweep(i)
    eng.sync()
    assert eng.ctx.rows_unfinished() == 0
    # D = 30: the 60,000-row entity's rows of at most 15 observations (nearly all) are drawn by the low-rank sampler (default-on
    # above 8,192 such rows: bdf_ctx_set_lowrank) -- the oracle dispatches the same way; the 1,000-row entity has none
    _compare(rd, *oracle_macau(O, rd, D, 21, 2, True, lowrank={0: 15} if D == 30 else None), tol=1e-6)
    eng.close()


def test_macau_with_side_information_and_lowrank_rows_match_oracle(B, O):
    """Macau at the shape side information is for (the reference's ChEMBL example, docs/index.md: thousands of compounds with a
    handful of measurements each and a feature vector): 12,000 x 300, 60,000 observations, dense entity features 12,000 x 24,
    D = 24.  The 12,000-row entity's rows of at most 12 observations -- nearly all -- are drawn by the low-rank sampler with
    PER-ROW prior means (mu + uhat_i: L' mu_i row by row); two whole native iterations (uhat, rows, hyperprior with the
    feature terms, beta by the direct solve, lambda_beta) against the oracle dispatching the same way."""
    from bdf_amd.engine import GibbsEngine
    rng = np.random.default_rng(24)
    N1, N2, D, numF, nnz = 12_000, 300, 24, 24, 60_000
    key = np.unique(rng.integers(0, N1 * N2, size=int(nnz * 1.02)))[:nnz]
    ids = np.stack([key // N2 + 1, key % N2 + 1], axis=1)
    F = rng.standard_normal((N1, numF))
    W = rng.standard_normal((numF, 3)) * 0.4
    vals = np.sum((F @ W)[ids[:, 0] - 1] * rng.standard_normal((N2, 3))[ids[:, 1] - 1], axis=1) + 0.3 * rng.standard_normal(nnz)
    rel = B.Relation((ids, vals), "r", [B.Entity("compounds", F=F), B.Entity("proteins")], dims=[N1, N2])
    B.setPrecision(rel, 2.0)
    rd = B.RelationData(rel)
    eng = GibbsEngine(rd, D, seed=31)
    assert eng.native and eng.lowrank_rows(0) > 11_000 and eng.lowrank_rows(1) == 0
    for i in (1, 2):
        eng.sweep(i)
    eng.sync()
    eng.sync_host_scalars()
    # what the library itself says it did (bdf_ctx_rows_dispatch), against the rule restated above and the relation's degrees
    deg = np.bincount(ids[:, 0] - 1, minlength=N1)
    d0, d1 = eng.rows_dispatch(0), eng.rows_dispatch(1)
    assert d0["lowrank"] == int((deg <= 12).sum()) == eng.lowrank_rows(0)
    assert d0["lowrank"] + d0["small"] + d0["col"] + d0["k1"] == N1 and d0["small"] == 0
    assert d1["lowrank"] == 0 and d1["col"] + d1["k1"] == N2 and eng.lowrank_rows(1) == 0
    _compare(rd, *oracle_macau(O, rd, D, 31, 2, True, lowrank={0: 12}), tol=1e-6)
    eng.close()


def test_c5_two_ranks_match_one():
    """C5's structure on two ranks -- a shared entity with two relations (3-mode + 2-mode) and binary sparse features, rows
    at internal positions, F's rows with them, noise keyed by the original ids, in-place exchange after every entity -- gives
    the chain of the single-process run (up to the summation order of the hyperprior's sums).  Two ranks on the box's one
    GPU: the exchange goes through the library's host transport (BDF_DIST_BACKEND=gloo)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "c5_ranks.py")
    one = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    port = str(29900 + os.getpid() % 90)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", port, tool], env=dict(os.environ, BDF_DIST_BACKEND="gloo", C5_CHUNKS="2"),
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    d2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["world"] == 2 and d1["world"] == 1
    assert d1["rmse"] < 0.6 * d1["value_std"]
    for k in ("rmse", "sample_norm", "beta_norm", "lambda_beta"):
        assert abs(d2[k] - d1[k]) <= 1e-6 * max(1.0, abs(d1[k])), (k, d1, d2)
    # the D conjugate-gradient solves are shared out over the ranks (parallel_matrix.jl:488-507): half the columns each, the
    # same iteration counts (all-gathered with the columns)
    assert d2["beta_columns_per_rank"] == 16 and d1["beta_columns_per_rank"] == 32 and d2["cg_iters"] == d1["cg_iters"]
    print("beta update: one rank %.3f ms, two ranks (same GPU, host transport) %.3f ms per call" % (d1["beta_update_ms"], d2["beta_update_ms"]))


@pytest.mark.parametrize("kind", ["dense", "csr"])
def test_relation_side_information_and_alpha_two_ranks_match_one(kind):
    """relation-level side information (sample_beta_rel) and alpha sampling with the rows shared out over two ranks: every rank
    holds one block of the observations (its rows of the relation's feature matrix, its observations as pairs); the squared
    errors, F'v and F'F are summed over the ranks in rank order (bdf_sum_ranks, bdf_sample_beta_rel_ranks), linear_values are
    gathered block by block.  Same chain as one process up to the summation order; the planted relation beta is recovered."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "relfeat_ranks.py")
    one = subprocess.run([sys.executable, tool, kind], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    port = str(29800 + os.getpid() % 90)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", port, tool, kind], env=dict(os.environ, BDF_DIST_BACKEND="gloo", RELFEAT_CHUNKS="2"),
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    d2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["world"] == 2 and d1["world"] == 1
    assert np.allclose(d1["beta_rel"], [1.0, -0.5, 2.0], atol=0.05), d1
    assert d1["rmse"] < 0.35 * d1["value_std"] and d1["alpha"] > 5.0, d1         # noise 0.2 -> precision ~ 25
    for k in ("rmse", "alpha", "alpha_5", "sample_norm", "linear_norm"):
        assert abs(d2[k] - d1[k]) <= 1e-6 * max(1.0, abs(d1[k])), (k, d1, d2)
    assert np.allclose(d2["beta_rel"], d1["beta_rel"], rtol=0, atol=1e-7), (d1, d2)


def test_rccl_one_rank_allgather(B, ctx):
    """librccl itself through bdf_comm (dlopen, ncclGetUniqueId, ncclCommInitRank, ncclAllGather on the communicator's stream,
    join): a ONE-rank communicator is all a 1-GPU box can hold, but it runs the whole call path; the N > 1 data movement is
    covered by the host transport (test_two_ranks_match_one) and, on CPU, by the gloo tests of the layout."""
    import ctypes as C
    import torch
    from bdf_amd import _lib
    from bdf_amd._lib import check, lib
    raw = (C.c_char * _lib.BDF_COMM_ID_BYTES)()
    check(lib().bdf_comm_unique_id(raw))
    assert any(b != b"\x00" for b in raw)
    comm = C.c_void_p()
    check(lib().bdf_comm_create(ctx.handle, 0, 1, raw.raw, C.byref(comm)))
    rank, world = C.c_int(-1), C.c_int(-1)
    check(lib().bdf_comm_size(comm, C.byref(rank), C.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    D, chunks, cmax = 8, 3, 50
    x = ctx.zeros(chunks * cmax, D)
    with torch.cuda.stream(ctx.stream):
        x.copy_(torch.arange(chunks * cmax * D, dtype=torch.float64, device=x.device).reshape(chunks * cmax, D))
    ref = x.clone()
    for c in range(chunks):
        check(lib().bdf_allgather_rows(ctx.handle, comm, D, chunks * cmax, _p(x), c, chunks))
    check(lib().bdf_allgather_join(ctx.handle, comm))
    ctx.sync()
    assert torch.equal(x, ref)
    assert lib().bdf_allgather_rows(ctx.handle, comm, D, chunks * cmax + 1, _p(x), 0, chunks) == -1      # rows not chunks x ranks x cmax
    check(lib().bdf_comm_destroy(comm))


@pytest.mark.parametrize("staged", [False, True])
def test_engine_communicator_transports(staged):
    """the engine's communicator over torch's RCCL process group (one rank: all a 1-GPU box holds): the library's own RCCL
    communicator, and the fall-back every rank agrees on when one of them cannot create it -- torch.distributed's all-gather
    behind the library's host transport (forced here with BDF_COMM_FORCE_STAGED)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BDF_DIST_BACKEND")}
    if staged:
        env["BDF_COMM_FORCE_STAGED"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "comm_transport_check.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("transport:")][-1]
    assert ("torch.distributed all-gather" in line) if staged else ("RCCL (ncclAllGather in place" in line), line
