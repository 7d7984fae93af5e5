"""GPU parity of the latent-row sampler (K1), RNG, hyperprior (K6) and prediction (K7) against the CPU oracle.

Every call goes through the C ABI (include/bdf.h).  Tolerances: integer / counter outputs bit-exact; the row system
(P_i, b_i) to 1e-12 relative; sampled rows to 1e-8 relative for the same normals (the oracle follows the reference's
inv + chol(covar), the device factors the precision once -- see k_sample_rows.hip); normals to 1e-13.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 1234


def _problem(rng, dims, nnz, D, empty_rows=True):
    n_modes = len(dims)
    ids = np.stack([rng.integers(1, d + 1, nnz) for d in dims], axis=1).astype(np.int64)
    if empty_rows and dims[0] > 3:
        ids[ids[:, 0] == 2, 0] = 1          # entity 2 of mode 1 has no observations
    vals = rng.standard_normal(nnz)
    facs = [rng.standard_normal((d, D)) for d in dims]
    A = rng.standard_normal((D, D))
    Lam = A @ A.T / D + np.eye(D)
    mu = rng.standard_normal(D)
    return ids, vals, facs, Lam, mu


def _dev_terms(B, ctx, rels):
    """rels: list of (DeviceRelation, mode0, alpha, mean, [factor tensors], linear tensor|None)"""
    from bdf_amd._lib import Term
    terms = (Term * len(rels))()
    for t, (dr, mode0, alpha, mean, facs, lin) in enumerate(rels):
        terms[t].rel = dr.handle
        terms[t].mode = mode0
        terms[t].alpha = alpha
        terms[t].mean_value = mean
        terms[t].linear_values = lin.data_ptr() if lin is not None else None
        for k, f in enumerate(facs):
            terms[t].factors[k] = f.data_ptr() if f is not None else None
    return terms


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, tag, out_t, shard=0, n_shards=1, pack_t=None):
    from bdf_amd._lib import check, lib
    check(lib().bdf_sample_rows(ctx.handle, D, N, len(terms), terms, _p(mu_t), int(mu_t.dim() == 2), _p(Lam_t), tag,
                                shard, n_shards, _p(out_t), _p(pack_t)))
    ctx.sync()


def test_philox_known_answer_and_stream(B, O, ctx):
    from bdf_amd._lib import check, lib
    out = (C.c_uint32 * 4)()
    # seed 0, sweep 0, purpose 0, entity 0, row 0, pair 0 == counter 0 / key 0: Random123 known answer
    c0 = B.Context(seed=0)
    c0.set_sweep(0)
    check(lib().bdf_philox(c0.handle, 0, 0, 0, 0, out))
    assert [hex(x) for x in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    c0.close()
    ctx.set_sweep(77)
    for (purpose, entity, row, pair) in [(1, 1, 0, 0), (1, 2, 123456789012, 31), (4, 0xabcdef, 7, 65535), (6, 3, 2 ** 40 + 5, 9)]:
        check(lib().bdf_philox(ctx.handle, purpose, entity, row, pair, out))
        assert list(out) == list(O.draw(SEED, 77, purpose, entity, row, pair))


@pytest.mark.parametrize("n", [1, 2, 7, 32, 64])
def test_normals_match_oracle(B, O, ctx, n):
    from bdf_amd._lib import check, lib
    ctx.set_sweep(5)
    rows = 300
    out = ctx.zeros(rows, n)
    check(lib().bdf_normals(ctx.handle, 1, 9, 1000, rows, n, _p(out)))
    ctx.sync()
    got = out.cpu().numpy()
    exp = np.stack([O.normals(SEED, 5, 1, 9, 1000 + r, n) for r in range(rows)])
    np.testing.assert_allclose(got, exp, rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("D", [1, 5, 10, 16, 17, 30, 32, 33, 64])
def test_row_system_and_sample_matrix(B, O, ctx, D):
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(D)
    dims = [57, 41]
    ids, vals, facs, Lam, mu = _problem(rng, dims, 900, D)
    idf = B.IndexedDF((ids, vals), dims)
    dr = B.DeviceRelation(ctx, idf)
    ft = [ctx.tensor(f) for f in facs]
    for mode0 in (0, 1):
        N = dims[mode0]
        alpha, mean = 1.7, 0.25
        fl = [None if k == mode0 else ft[k] for k in range(2)]
        terms = _dev_terms(B, ctx, [(dr, mode0, alpha, mean, fl, None)])
        Lam_t, mu_t = ctx.tensor(Lam), ctx.tensor(mu)
        P_t, b_t = ctx.zeros(N, D, D), ctx.zeros(N, D)
        check(lib().bdf_row_system(ctx.handle, D, N, 1, terms, _p(mu_t), 0, _p(Lam_t), _p(P_t), _p(b_t)))
        ctx.sync()
        P, b = P_t.cpu().numpy(), b_t.cpu().numpy()
        ot = O.Term(ids, vals, dims, mode0, alpha, mean, [None if k == mode0 else facs[k] for k in range(2)])
        for row in range(N):
            Pe, be = O.row_system(D, [ot], row, mu, Lam)
            np.testing.assert_allclose(P[row].T, Pe, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(b[row], be, rtol=1e-12, atol=1e-12)
        # full draw, same stream
        ctx.set_sweep(3)
        out_t = ctx.zeros(N, D)
        _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 11, out_t)
        exp = O.sample_rows(D, N, [ot], mu, Lam, SEED, 3, 11)
        np.testing.assert_allclose(out_t.cpu().numpy(), exp, rtol=1e-8, atol=1e-9)
    dr.close()


@pytest.mark.parametrize("D", [20, 32, 64])
def test_per_row_prior_means_of_many_rows(B, O, ctx, D):
    """Lambda mu_i for an entity of >= 4,096 rows comes from k_prior_rows (round 6: one thread per element, its row of Lambda in
    registers) instead of k_prior (eight lanes and a butterfly per element) -- the same sums in the same order: b_i of the first 4,000
    rows is BIT-IDENTICAL between a 4,500-row entity (new kernel) and a 4,000-row one (k_prior) with the same observations, equals
    Lambda mu_i + the data term to 1e-12, and the drawn rows match the oracle (macau.jl:104, sampling.jl:200-212)."""
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(900 + D)
    N_big, N_small, M = 4500, 4000, 30
    ids, vals, facs, Lam, mu = _problem(rng, [N_small, M], 3000, D)
    mu_rows = rng.standard_normal((N_big, D))
    f1 = ctx.tensor(facs[1])
    Lam_t = ctx.tensor(Lam)
    b = {}
    for N in (N_big, N_small):
        dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), [N, M]))
        terms = _dev_terms(B, ctx, [(dr, 0, 1.3, 0.2, [None, f1], None)])
        mu_t = ctx.tensor(mu_rows[:N])
        P_t, b_t = ctx.zeros(N, D, D), ctx.zeros(N, D)
        check(lib().bdf_row_system(ctx.handle, D, N, 1, terms, _p(mu_t), 1, _p(Lam_t), _p(P_t), _p(b_t)))
        ctx.sync()
        b[N] = b_t.cpu().numpy()
        if N == N_big:
            ctx.set_sweep(4)
            out_t = ctx.zeros(N, D)
            _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 6, out_t)
            ot = O.Term(ids, vals, [N, M], 0, 1.3, 0.2, [None, facs[1]])
            exp = O.sample_rows(D, N, [ot], mu_rows, Lam, SEED, 4, 6)
            np.testing.assert_allclose(out_t.cpu().numpy(), exp, rtol=1e-8, atol=1e-9)
        dr.close()
    assert np.array_equal(b[N_big][:N_small], b[N_small])
    empty = np.setdiff1d(np.arange(N_big), ids[:, 0] - 1)
    np.testing.assert_allclose(b[N_big][empty], (mu_rows @ Lam.T)[empty], rtol=1e-12, atol=1e-12)


def test_rows_tensor_multi_relation_mu_matrix_linear(B, O, ctx):
    """3-mode relation (Hadamard gather, sampling.jl:215-234) + a 2-mode relation sharing the entity
    (sum over relations, :270-283), per-row prior mean (macau.jl:104) and linear_values (:273)"""
    D = 12
    rng = np.random.default_rng(5)
    dimsA, dimsB = [23, 9, 6], [23, 14]
    idsA, valsA, facsA, Lam, mu = _problem(rng, dimsA, 700, D)
    idsB, valsB, facsB, _, _ = _problem(rng, dimsB, 300, D)
    facsB[0] = facsA[0]
    linB = rng.standard_normal(len(valsB))
    mu_mat = rng.standard_normal((23, D))
    idfA, idfB = B.IndexedDF((idsA, valsA), dimsA), B.IndexedDF((idsB, valsB), dimsB)
    drA, drB = B.DeviceRelation(ctx, idfA), B.DeviceRelation(ctx, idfB)
    fA = [ctx.tensor(f) for f in facsA]
    fB = [fA[0], ctx.tensor(facsB[1])]
    lin_t = ctx.tensor(linB)
    terms = _dev_terms(B, ctx, [(drA, 0, 2.0, 0.1, [None, fA[1], fA[2]], None), (drB, 0, 0.7, -0.3, [None, fB[1]], lin_t)])
    Lam_t, mu_t = ctx.tensor(Lam), ctx.tensor(mu_mat)
    ctx.set_sweep(9)
    out_t = ctx.zeros(23, D)
    _run_rows(B, ctx, D, 23, terms, mu_t, Lam_t, 2, out_t)
    oA = O.Term(idsA, valsA, dimsA, 0, 2.0, 0.1, [None, facsA[1], facsA[2]])
    oB = O.Term(idsB, valsB, dimsB, 0, 0.7, -0.3, [None, facsB[1]], linear_values=linB)
    exp = O.sample_rows(D, 23, [oA, oB], mu_mat, Lam, SEED, 9, 2)
    np.testing.assert_allclose(out_t.cpu().numpy(), exp, rtol=1e-8, atol=1e-9)
    # middle mode of the tensor
    terms = _dev_terms(B, ctx, [(drA, 1, 2.0, 0.1, [fA[0], None, fA[2]], None)])
    out_t = ctx.zeros(9, D)
    mu1_t = ctx.tensor(mu)
    _run_rows(B, ctx, D, 9, terms, mu1_t, Lam_t, 3, out_t)
    oA1 = O.Term(idsA, valsA, dimsA, 1, 2.0, 0.1, [facsA[0], None, facsA[2]])
    exp = O.sample_rows(D, 9, [oA1], mu, Lam, SEED, 9, 3)
    np.testing.assert_allclose(out_t.cpu().numpy(), exp, rtol=1e-8, atol=1e-9)
    drA.close(); drB.close()


def test_shard_writes_only_its_rows(B, O, ctx):
    """(shard, n_shards) = positions shard::n_shards of the degree-descending order (the reference's i:P:N deal,
    sampling.jl:154); the union of the shards equals the unsharded result"""
    D = 8
    rng = np.random.default_rng(8)
    dims = [40, 30]
    ids, vals, facs, Lam, mu = _problem(rng, dims, 500, D)
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    ft = [ctx.tensor(f) for f in facs]
    terms = _dev_terms(B, ctx, [(dr, 0, 1.0, 0.0, [None, ft[1]], None)])
    ctx.set_sweep(1)
    mu_t, Lam_t = ctx.tensor(mu), ctx.tensor(Lam)
    exp = O.sample_rows(D, 40, [O.Term(ids, vals, dims, 0, 1.0, 0.0, [None, facs[1]])], mu, Lam, SEED, 1, 1)
    order = dr.order(0)
    counts = np.bincount(ids[:, 0] - 1, minlength=40)
    assert sorted(order.tolist()) == list(range(40))
    assert np.all(np.diff(counts[order]) <= 0)
    for n_shards in (2, 3):
        for shard in range(n_shards):
            out_t = ctx.zeros(40, D) + 777.0
            _run_rows(B, ctx, D, 40, terms, mu_t, Lam_t, 1, out_t, shard, n_shards)
            got = out_t.cpu().numpy()
            mask = np.zeros(40, dtype=bool)
            mask[order[shard::n_shards]] = True
            np.testing.assert_allclose(got[mask], exp[mask], rtol=1e-8, atol=1e-9)
            assert np.all(got[~mask] == 777.0)
    dr.close()


@pytest.mark.parametrize("item,piece", [(8, None), (16, None), (1000, None), (64, 24), (192, 128), (100, 100)])
def test_item_size_does_not_change_results(B, O, ctx, item, piece):
    """rows longer than the item size are split into pieces (of the piece size) over wavefronts and re-assembled in slot order"""
    D = 32
    rng = np.random.default_rng(77)
    dims = [12, 50]
    ids, vals, facs, Lam, mu = _problem(rng, dims, 2500, D, empty_rows=True)
    c2 = B.Context(seed=SEED)
    c2.set_item_size(item)
    if piece is not None:
        c2.set_piece_size(piece)
    dr = B.DeviceRelation(c2, B.IndexedDF((ids, vals), dims))
    ft = [c2.tensor(f) for f in facs]
    terms = _dev_terms(B, c2, [(dr, 0, 1.3, 0.2, [None, ft[1]], None)])
    c2.set_sweep(4)
    mu_t, Lam_t = c2.tensor(mu), c2.tensor(Lam)
    out_t = c2.zeros(12, D)
    _run_rows(B, c2, D, 12, terms, mu_t, Lam_t, 6, out_t)
    exp = O.sample_rows(D, 12, [O.Term(ids, vals, dims, 0, 1.3, 0.2, [None, facs[1]])], mu, Lam, SEED, 4, 6)
    np.testing.assert_allclose(out_t.cpu().numpy(), exp, rtol=1e-8, atol=1e-9)
    dr.close()
    c2.close()


def test_piece_cap(B, O, ctx):
    """a row with more than 64 pieces' worth of observations is cut into 64 (longer) pieces: two rows of ~1250 observations at
    item size 8 would be 157 pieces each"""
    D = 16
    rng = np.random.default_rng(78)
    dims = [2, 3000]
    ids, vals, facs, Lam, mu = _problem(rng, dims, 2500, D, empty_rows=False)
    c2 = B.Context(seed=SEED)
    c2.set_item_size(8)
    dr = B.DeviceRelation(c2, B.IndexedDF((ids, vals), dims))
    ft = [c2.tensor(f) for f in facs]
    terms = _dev_terms(B, c2, [(dr, 0, 0.7, -0.1, [None, ft[1]], None)])
    c2.set_sweep(2)
    out_t = c2.zeros(2, D)
    _run_rows(B, c2, D, 2, terms, c2.tensor(mu), c2.tensor(Lam), 3, out_t)
    exp = O.sample_rows(D, 2, [O.Term(ids, vals, dims, 0, 0.7, -0.1, [None, facs[1]])], mu, Lam, SEED, 2, 3)
    np.testing.assert_allclose(out_t.cpu().numpy(), exp, rtol=1e-8, atol=1e-9)
    assert c2.rows_unfinished() == 0
    dr.close()
    c2.close()


def test_row_moments(B, O, ctx):
    """sampled moments of one row over many sweeps: mean within 5 sigma/sqrt(n), covariance within 5% (Frobenius)"""
    D = 6
    rng = np.random.default_rng(21)
    dims = [4, 12]
    ids, vals, facs, Lam, mu = _problem(rng, dims, 40, D, empty_rows=False)
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    ft = [ctx.tensor(f) for f in facs]
    terms = _dev_terms(B, ctx, [(dr, 0, 2.0, 0.1, [None, ft[1]], None)])
    Lam_t, mu_t = ctx.tensor(Lam), ctx.tensor(mu)
    n = 20000
    draws = np.zeros((n, D))
    out_t = ctx.zeros(4, D)
    from bdf_amd._lib import check, lib
    for s in range(n):
        ctx.set_sweep(s + 1)
        check(lib().bdf_sample_rows(ctx.handle, D, 4, 1, terms, _p(mu_t), 0, _p(Lam_t), 1, 0, 1, _p(out_t), None))
        draws[s] = out_t[1].cpu().numpy()
    P, b = O.row_system(D, [O.Term(ids, vals, dims, 0, 2.0, 0.1, [None, facs[1]])], 1, mu, Lam)
    cov = np.linalg.inv(P)
    mean = cov @ b
    se = np.sqrt(np.diag(cov) / n)
    assert np.all(np.abs(draws.mean(0) - mean) < 5 * se)
    emp = np.cov(draws.T)
    assert np.linalg.norm(emp - cov) / np.linalg.norm(cov) < 0.05
    dr.close()


def test_not_positive_definite_is_reported(B, ctx):
    D = 4
    rng = np.random.default_rng(3)
    dims = [5, 5]
    ids, vals, facs, Lam, mu = _problem(rng, dims, 20, D, empty_rows=False)
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    ft = [ctx.tensor(f) for f in facs]
    terms = _dev_terms(B, ctx, [(dr, 0, 1.0, 0.0, [None, ft[1]], None)])
    bad_t, mu_t = ctx.tensor(-np.eye(D) * 1e6), ctx.tensor(mu)
    out_t = ctx.zeros(5, D)
    from bdf_amd._lib import check, lib
    check(lib().bdf_sample_rows(ctx.handle, D, 5, 1, terms, _p(mu_t), 0, _p(bad_t), 1, 0, 1, _p(out_t), None))
    with pytest.raises(B.NotPositiveDefinite):
        ctx.sync()
    ctx.sync()      # flag is cleared
    dr.close()


def test_argument_errors(B, ctx):
    from bdf_amd._lib import lib
    D = 4
    dims = [5, 6]
    ids = np.array([[1, 1], [5, 6]])
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, np.ones(2)), dims))
    f1, f2 = ctx.zeros(5, D), ctx.zeros(6, D)
    terms = _dev_terms(B, ctx, [(dr, 0, 1.0, 0.0, [None, f2], None)])
    mu, Lam = ctx.zeros(D), ctx.tensor(np.eye(D))
    # entity count disagrees with the relation (ArgumentError, RelationData.jl:399)
    assert lib().bdf_sample_rows(ctx.handle, D, 7, 1, terms, _p(mu), 0, _p(Lam), 1, 0, 1, _p(f1), None) == -1
    # num_latent out of range
    assert lib().bdf_sample_rows(ctx.handle, 65, 5, 1, terms, _p(mu), 0, _p(Lam), 1, 0, 1, _p(f1), None) == -1
    # output aliasing a gathered factor
    assert lib().bdf_sample_rows(ctx.handle, D, 5, 1, terms, _p(mu), 0, _p(Lam), 1, 0, 1, _p(f2), None) == -1
    # shard outside 0..n_shards-1
    assert lib().bdf_sample_rows(ctx.handle, D, 5, 1, terms, _p(mu), 0, _p(Lam), 1, 2, 2, _p(f1), None) == -1
    with pytest.raises(B.BoundsError):
        B.DeviceRelation(ctx, type("X", (), {"dims": [2, 2], "values": np.ones(1), "ids": np.asfortranarray(np.array([[3, 1]])),
                                             "nnz": lambda self: 1})())
    # the low-rank sampler's longest row: -1 (default), 0 (off), 1 .. 32 (above 16 only D > 32 has a kernel for; clamped otherwise)
    assert lib().bdf_ctx_set_lowrank(ctx.handle, 33, 0) == -1 and lib().bdf_ctx_set_lowrank(ctx.handle, -2, 0) == -1
    assert lib().bdf_ctx_set_lowrank(ctx.handle, 32, 0) == 0 and lib().bdf_ctx_set_lowrank(ctx.handle, -1, 8192) == 0
    # no row launch under a tag yet: the dispatch report says so
    assert ctx.rows_dispatch(123456) is None
    dr.close()


def test_device_index_is_the_reference_index(B, O, ctx):
    rng = np.random.default_rng(0)
    dims = [13, 7, 5]
    ids = np.stack([rng.integers(1, d + 1, 400) for d in dims], axis=1)
    idf = B.IndexedDF((ids, rng.standard_normal(400)), dims)
    dr = B.DeviceRelation(ctx, idf)
    ref = O.index_build(ids, dims)
    for m in range(3):
        rp, ri = dr.index(m)
        assert np.array_equal(rp, ref[m][0]) and np.array_equal(ri, ref[m][1])
    assert abs(dr.value_mean() - idf.valueMean()) < 1e-15
    dr.close()


@pytest.mark.parametrize("D,N,with_uhat", [(3, 10, False), (10, 1000, True), (32, 6040, False), (64, 777, True)])
def test_hyper_sums_and_normal_wishart(B, O, ctx, D, N, with_uhat):
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(D + N)
    S = rng.standard_normal((N, D)) * 0.7 + rng.standard_normal(D)
    uh = rng.standard_normal((N, D)) * 0.1 if with_uhat else None
    U = S - uh if with_uhat else S
    S_t = ctx.tensor(S)
    uh_t = ctx.tensor(uh) if with_uhat else None
    sumU, UUt = ctx.zeros(D), ctx.zeros(D, D)
    check(lib().bdf_hyper_sums(ctx.handle, D, N, _p(S_t), _p(uh_t), _p(sumU), _p(UUt)))
    ctx.sync()
    np.testing.assert_allclose(sumU.cpu().numpy(), U.sum(0), rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(UUt.cpu().numpy(), U.T @ U, rtol=1e-12, atol=1e-10)
    # Normal-Wishart: parameters and draw against the oracle on the same streams
    mu0 = rng.standard_normal(D) * 0.1
    A = rng.standard_normal((D, D))
    Tinv = A @ A.T / D + np.eye(D)
    b0, nu = 2.0, D + 3.0
    ctx.set_sweep(12)
    mu_t, Lam_t, par_t = ctx.zeros(D), ctx.zeros(D, D), ctx.zeros(D + D * D)
    mu0_t, Tinv_t = ctx.tensor(mu0), ctx.tensor(Tinv)      # keep alive until the kernel has run
    check(lib().bdf_hyper_sample(ctx.handle, D, N, _p(sumU), _p(UUt), _p(mu0_t), b0, _p(Tinv_t), nu, 5,
                                 _p(mu_t), _p(Lam_t), _p(par_t), None, None))
    ctx.sync()
    mu_N, beta_N, T_N, nu_N = O.hyper_params(U, mu0, b0, Tinv, nu)
    par = par_t.cpu().numpy()
    np.testing.assert_allclose(par[:D], mu_N, rtol=1e-11, atol=1e-12)
    W = par[D:].reshape(D, D)
    np.testing.assert_allclose(np.triu(W), np.triu(np.linalg.inv(T_N)), rtol=1e-8, atol=1e-8)
    mu_e, Lam_e = O.hyper_draw(mu_N, beta_N, T_N, nu_N, SEED, 12, 5)
    np.testing.assert_allclose(Lam_t.cpu().numpy(), Lam_e, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(mu_t.cpu().numpy(), mu_e, rtol=1e-7, atol=1e-9)
    # the random part drawn ahead of time (bdf_hyper_draws) gives the same bits
    draws_t, mu2_t, Lam2_t = ctx.zeros(D * D + D), ctx.zeros(D), ctx.zeros(D, D)
    check(lib().bdf_hyper_draws(ctx.handle, D, N, nu, 5, _p(draws_t)))
    check(lib().bdf_hyper_sample(ctx.handle, D, N, _p(sumU), _p(UUt), _p(mu0_t), b0, _p(Tinv_t), nu, 5,
                                 _p(mu2_t), _p(Lam2_t), None, None, _p(draws_t)))
    ctx.sync()
    assert np.array_equal(mu2_t.cpu().numpy(), mu_t.cpu().numpy())
    assert np.array_equal(Lam2_t.cpu().numpy(), Lam_t.cpu().numpy())


@pytest.mark.parametrize("D", [5, 16, 32, 40])
def test_prior_pack_equals_prelaunch(B, O, ctx, D):
    """bdf_hyper_sample's prior pack (Lambda mu and the accumulator-layout image of Lambda) gives bdf_sample_rows bit for bit
    what its own pre-launch derives from (mu, Lambda)"""
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(100 + D)
    N = 300
    S_t = ctx.tensor(rng.standard_normal((N, D)))
    sumU, UUt = ctx.zeros(D), ctx.zeros(D, D)
    check(lib().bdf_hyper_sums(ctx.handle, D, N, _p(S_t), None, _p(sumU), _p(UUt)))
    mu0_t, Tinv_t = ctx.zeros(D), ctx.tensor(np.eye(D))
    mu_t, Lam_t = ctx.zeros(D), ctx.zeros(D, D)
    pack_t = ctx.zeros(lib().bdf_prior_pack_doubles(D))
    ctx.set_sweep(3)
    check(lib().bdf_hyper_sample(ctx.handle, D, N, _p(sumU), _p(UUt), _p(mu0_t), 2.0, _p(Tinv_t), float(D), 9,
                                 _p(mu_t), _p(Lam_t), None, _p(pack_t), None))
    ctx.sync()
    Lam, mu = Lam_t.cpu().numpy(), mu_t.cpu().numpy()
    np.testing.assert_allclose(pack_t.cpu().numpy()[:D], Lam @ mu, rtol=1e-12, atol=1e-12)
    dims = [40, 30]
    ids, vals, facs, _, _ = _problem(rng, dims, 700, D)
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    ft = [ctx.tensor(f) for f in facs]
    terms = _dev_terms(B, ctx, [(dr, 0, 1.7, float(vals.mean()), [None, ft[1]], None)])
    out_a, out_b = ctx.zeros(dims[0], D), ctx.zeros(dims[0], D)
    ctx.set_sweep(4)
    _run_rows(B, ctx, D, dims[0], terms, mu_t, Lam_t, 2, out_a)
    _run_rows(B, ctx, D, dims[0], terms, mu_t, Lam_t, 2, out_b, pack_t=pack_t)
    assert np.array_equal(out_a.cpu().numpy(), out_b.cpu().numpy())
    dr.close()


def test_predict_and_running_mean(B, O, ctx):
    D = 10
    rng = np.random.default_rng(4)
    dims = [30, 20, 4]
    n = 1000
    ids = np.stack([rng.integers(1, d + 1, n) for d in dims], axis=1)
    y = rng.standard_normal(n) + 3
    pairs = B.DevicePairs(ctx, ids, y)
    clamp, cut = [2.0, 4.0], 3.0
    avg = sq = None
    for it, phase in enumerate([0, 0, 1, 2, 2, 2]):
        facs = [rng.standard_normal((d, D)) * 0.5 for d in dims]
        ft = [ctx.tensor(f) for f in facs]
        p = O.predict(ids, facs, 3.0)
        got = pairs.predict(D, ft, 3.0).cpu().numpy()
        np.testing.assert_allclose(got, p, rtol=1e-12, atol=1e-12)
        stats = pairs.update(D, ft, 3.0, phase, clamp, cut).cpu().numpy()
        if phase == 0:
            avg = p
        elif phase == 1:
            avg, sq, cnt = p, p ** 2, 1
        else:
            avg = (cnt * avg + p) / (cnt + 1); sq = sq + p ** 2; cnt += 1
        ca, cp = np.clip(avg, *clamp), np.clip(p, *clamp)
        exp = [np.sum((y - ca) ** 2), np.sum((y - cp) ** 2), np.sum((y < cut) == (avg < cut)), np.sum((y < cut) == (p < cut))]
        np.testing.assert_allclose(stats, exp, rtol=1e-10)
    a, s = pairs.state()
    np.testing.assert_allclose(a, avg, rtol=1e-12)
    np.testing.assert_allclose(s, sq, rtol=1e-12)
    pairs.close()


@pytest.mark.parametrize("n_modes,D", [(2, 3), (2, 8), (2, 32), (2, 33), (2, 64), (3, 5), (3, 16), (3, 64), (4, 12), (4, 64), (4, 7)])
def test_predict_every_kernel_variant_ragged_counts(B, O, ctx, n_modes, D):
    """the prediction kernels by mode count and row width (scalar path for D not a multiple of 4, one or two 32-byte pieces per
    lane otherwise; the run kernel for sorted two-mode pairs up to D = 32) at pair counts around the 8-pair trips and 16-pair runs"""
    rng = np.random.default_rng(100 * n_modes + D)
    dims = [13, 9, 4, 3][:n_modes]
    facs = [rng.standard_normal((d, D)) * 0.6 for d in dims]
    ft = [ctx.tensor(f) for f in facs]
    clamp, cut = [2.0, 4.0], 3.0
    for n in (1, 7, 8, 9, 15, 16, 17, 255, 257):
        ids = np.stack([rng.integers(1, d + 1, n) for d in dims], axis=1)
        y = rng.standard_normal(n) + 3
        exp = O.predict(ids, facs, 2.5)
        variants = [B.DevicePairs(ctx, ids, y)]
        if n_modes == 2:
            variants += [B.DevicePairs(ctx, ids, y).sort(0), B.DevicePairs(ctx, ids, y).sort(1)]
        for pairs in variants:
            np.testing.assert_allclose(pairs.predict(D, ft, 2.5).cpu().numpy(), exp, rtol=1e-12, atol=1e-12)
            for phase in (0, 1, 2):
                stats = pairs.update(D, ft, 2.5, phase, clamp, cut).cpu().numpy()
            ca = np.clip(exp, *clamp)          # the same factors every time: running mean == the prediction
            np.testing.assert_allclose(stats, [np.sum((y - ca) ** 2), np.sum((y - ca) ** 2), np.sum((y < cut) == (exp < cut)),
                                               np.sum((y < cut) == (exp < cut))], rtol=1e-10, atol=1e-12)
            a, sq = pairs.state()                 # in the caller's order
            np.testing.assert_allclose(a, exp, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(sq, 2 * exp ** 2, rtol=1e-12, atol=1e-12)
            pairs.close()


@pytest.mark.parametrize("dims,D", [([37, 21], 32), ([6040 // 40, 99], 10), ([19, 5, 7], 12), ([9, 4, 3, 5], 64), ([1, 1], 1), ([33, 0], 8)])
def test_predict_all_cells(B, ctx, dims, D):
    """bdf_predict_all = pred_all(r) (sampling.jl:91-97): udot over every cell + mean_value, the last mode fastest; sizes that are
    no multiple of the 16 x 16 tile, two to four modes, an empty mode"""
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(sum(dims) + D)
    facs = [rng.standard_normal((d, D)) * 0.7 for d in dims]
    ft = [ctx.tensor(f) if f.size else ctx.zeros(1, D) for f in facs]
    out = ctx.zeros(*[max(d, 1) for d in dims])
    out.fill_(-7.0)
    fp = (C.c_void_p * len(dims))(*[t.data_ptr() for t in ft])
    check(lib().bdf_predict_all(ctx.handle, len(dims), (C.c_int64 * len(dims))(*dims), D, fp, 0.35, _p(out)))
    ctx.sync()
    if 0 in dims:
        assert float(out.min()) == -7.0                     # nothing written
        return
    letters = "abcd"[:len(dims)]
    exp = np.einsum(",".join(f"{c}z" for c in letters) + "->" + letters, *facs) + 0.35
    np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=1e-12, atol=1e-12)


def test_hyper_sums_large_entity(B, ctx):
    """more than 2048 x 128 rows: every workgroup of the sums kernel takes several 128-row chunks"""
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(77)
    D, N = 8, 2048 * 128 * 2 + 77
    S = rng.standard_normal((N, D))
    S_t = ctx.tensor(S)
    sumU, UUt = ctx.zeros(D), ctx.zeros(D, D)
    check(lib().bdf_hyper_sums(ctx.handle, D, N, _p(S_t), None, _p(sumU), _p(UUt)))
    ctx.sync()
    np.testing.assert_allclose(sumU.cpu().numpy(), S.sum(0), rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(UUt.cpu().numpy(), S.T @ S, rtol=1e-11, atol=1e-8)


@pytest.mark.parametrize("mode", ["general", "wide", "general_kernel"])
def test_gather_paths_agree(B, ctx, mode):
    """the lean gather (32-bit offsets), its 64-bit variant for factor matrices of 4 GiB and more, the general path, and the
    general kernel variant in place of the two-mode one (BDF_K1_GENERAL_KERNEL) give the same rows (the hooks are read
    once per process: run the forced path in a child)"""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import ctypes as C, numpy as np, sys
        sys.path.insert(0, %r)
        import bdf_amd as B
        from bdf_amd._lib import Term, check, lib
        rng = np.random.default_rng(5)
        D, dims = 48, [70, 50]
        ids = np.stack([rng.integers(1, d + 1, 3000) for d in dims], axis=1).astype(np.int64)
        vals = rng.standard_normal(3000)
        ctx = B.Context(seed=11)
        dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
        ft = [ctx.tensor(rng.standard_normal((d, D))) for d in dims]
        A = rng.standard_normal((D, D)); Lam = ctx.tensor(A @ A.T / D + np.eye(D)); mu = ctx.tensor(rng.standard_normal(D))
        terms = (Term * 1)()
        terms[0].rel = dr.handle; terms[0].mode = 0; terms[0].alpha = 1.3; terms[0].mean_value = 0.1
        terms[0].factors[1] = ft[1].data_ptr()
        out = ctx.zeros(dims[0], D)
        ctx.set_sweep(2)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(lib().bdf_sample_rows(ctx.handle, D, dims[0], 1, terms, p(mu), 0, p(Lam), 1, 0, 1, p(out), None))
        ctx.sync()
        np.save(sys.argv[1], out.cpu().numpy())
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    outs = {}
    for m in ("", mode):
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "o.npy")
            env = dict(os.environ)
            env.pop("BDF_GATHER", None)
            env.pop("BDF_K1_GENERAL_KERNEL", None)
            if m == "general_kernel":
                env["BDF_K1_GENERAL_KERNEL"] = "1"
            elif m:
                env["BDF_GATHER"] = m
            subprocess.run([sys.executable, "-c", code, f], check=True, env=env, timeout=300)
            outs[m] = np.load(f)
    np.testing.assert_allclose(outs[mode], outs[""], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("D,n_values", [(32, 5), (20, 10), (64, 5), (10, 32), (32, 33)])
def test_coded_values_variant_is_bit_identical(B, O, ctx, D, n_values):
    """a two-mode relation whose values are a few distinct numbers (ratings) takes the row kernel's coded variant -- id and
    8-bit value code in one word, the value from a table in LDS, seven resident waves at D <= 32; up to 32 distinct values,
    33 fall back to the plain two-mode variant.  Same arithmetic: the rows equal the uncoded variant's (BDF_K1_NO_CODED, read
    once per process: a child) to the last bit, and the oracle's to tolerance.  Rows of 0 .. 900 observations: whole rows,
    ragged last trips, split rows.  At 16 < D <= 32 the same pair once more through K1c (k_rows_col.hip, pieces of 48: the packed
    word and the table in global memory against ids and values): bit-identical to each other, equal to the wave-per-row kernel to
    rounding."""
    import subprocess, sys, textwrap, tempfile
    code = textwrap.dedent('''
        import ctypes as C, numpy as np, sys
        sys.path.insert(0, %r)
        import bdf_amd as B
        from bdf_amd._lib import Term, check, lib
        D, n_values = int(sys.argv[2]), int(sys.argv[3])
        rng = np.random.default_rng(7 + D)
        dims = [300, 120]
        deg = np.minimum((rng.pareto(1.2, dims[0]) * 12).astype(int), 900)
        deg[:3] = [0, 1, 900]
        rows = np.repeat(np.arange(1, dims[0] + 1), deg)
        ids = np.stack([rows, rng.integers(1, dims[1] + 1, len(rows))], axis=1).astype(np.int64)
        table = np.sort(rng.choice(np.arange(-20, 60) * 0.25, n_values, replace=False))
        vals = table[rng.integers(0, n_values, len(rows))]
        vals[:n_values] = table                              # every value occurs
        ctx = B.Context(seed=11)
        ctx.set_item_size(160)
        ctx.set_piece_size(128)
        ctx.set_col_rows(int(sys.argv[4]))
        dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
        ft = [ctx.tensor(rng.standard_normal((d, D)) * 0.4) for d in dims]
        A = rng.standard_normal((D, D)); Lam = ctx.tensor(A @ A.T / D + np.eye(D)); mu = ctx.tensor(rng.standard_normal(D))
        outs = []
        for mode in (0, 1):
            terms = (Term * 1)()
            terms[0].rel = dr.handle; terms[0].mode = mode; terms[0].alpha = 1.3; terms[0].mean_value = float(vals.mean())
            terms[0].factors[1 - mode] = ft[1 - mode].data_ptr()
            out = ctx.zeros(dims[mode], D)
            ctx.set_sweep(2)
            p = lambda t: C.c_void_p(t.data_ptr())
            check(lib().bdf_sample_rows(ctx.handle, D, dims[mode], 1, terms, p(mu), 0, p(Lam), 1 + mode, 0, 1, p(out), None))
            ctx.sync()
            outs.append(out.cpu().numpy())
        np.savez(sys.argv[1], u=outs[0], v=outs[1], ids=ids, vals=vals, f0=ft[0].cpu().numpy(), f1=ft[1].cpu().numpy(),
                 Lam=Lam.cpu().numpy(), mu=mu.cpu().numpy())
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for variant in ("coded", "uncoded") + (("col_coded", "col_uncoded") if 16 < D <= 32 else ()):
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "o.npz")
            env = dict(os.environ)
            env.pop("BDF_K1_NO_CODED", None); env.pop("BDF_NO_CODES", None)
            if variant.endswith("uncoded"):
                env["BDF_K1_NO_CODED"] = "1"
            subprocess.run([sys.executable, "-c", code, f, str(D), str(n_values), "48" if variant.startswith("col") else "0"], check=True,
                           env=env, timeout=300)
            got[variant] = dict(np.load(f))
    for k in ("u", "v"):
        assert np.array_equal(got["coded"][k], got["uncoded"][k]), k
        if "col_coded" in got:        # K1c: the same LDL' factorisation in another layout -- equal to rounding, not to the last bit
            assert np.array_equal(got["col_coded"][k], got["col_uncoded"][k]), k
            np.testing.assert_allclose(got["col_coded"][k], got["coded"][k], rtol=1e-9, atol=1e-11)
    g = got["coded"]
    dims = [300, 120]
    idx = O.index_build(g["ids"], dims)
    for mode, key in ((0, "u"), (1, "v")):
        t = O.Term(g["ids"], g["vals"], dims, mode, 1.3, float(g["vals"].mean()), [None if k == mode else g["f%d" % k] for k in (0, 1)], index=idx)
        exp = O.sample_rows(D, dims[mode], [t], g["mu"], g["Lam"], 11, 2, 1 + mode)
        np.testing.assert_allclose(g[key], exp, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("D,mode", [(12, 0), (32, 1), (32, 0), (10, 1), (4, 0)])
def test_sorted_pairs_keep_the_callers_order(B, O, ctx, D, mode):
    """bdf_pairs_sort: same predictions, running means and statistics as the unsorted pairs, in the caller's order (D a
    multiple of 4 up to 32: the run kernel that keeps the sorted mode's row in registers; otherwise the general kernel)"""
    rng = np.random.default_rng(14 + D + mode)
    dims = [40, 25]
    n = 3000
    ids = np.stack([rng.integers(1, d + 1, n) for d in dims], axis=1)
    y = rng.standard_normal(n) + 3
    facs = [rng.standard_normal((d, D)) * 0.5 for d in dims]
    ft = [ctx.tensor(f) for f in facs]
    plain, srt = B.DevicePairs(ctx, ids, y), B.DevicePairs(ctx, ids, y).sort(mode)
    np.testing.assert_array_equal(srt.predict(D, ft, 0.4).cpu().numpy(), plain.predict(D, ft, 0.4).cpu().numpy())
    for phase in (0, 1, 2, 2):
        for q in ft:
            q.mul_(0.9)
        s1 = plain.update(D, ft, 0.4, phase, [2.0, 4.0], 3.0).cpu().numpy().copy()
        s2 = srt.update(D, ft, 0.4, phase, [2.0, 4.0], 3.0).cpu().numpy().copy()
        np.testing.assert_allclose(s2, s1, rtol=1e-12)
    a1, q1 = plain.state()
    a2, q2 = srt.state()
    np.testing.assert_array_equal(a2, a1)
    np.testing.assert_array_equal(q2, q1)
    plain.close(); srt.close()


def test_sample_users_blocked(B, O, ctx):
    """Block / sample_users_blocked (sampling.jl:236-249): the users of a block share one covariance;
    covar = inv(Lambda + alpha MM MM'), mu = covar (alpha MM Yma + Lambda mu_u), sample = chol(covar)' z + mu"""
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(31)
    D, M = 12, 40
    sample_mt = rng.standard_normal((D, M))
    vx = np.array([3, 7, 8, 20, 33, 40])
    ux = np.array([5, 2, 9, 11])
    Yma = rng.standard_normal((len(vx), len(ux)))
    A = rng.standard_normal((D, D))
    Lam, mu, alpha = A @ A.T / D + np.eye(D), rng.standard_normal(D), 1.7
    blk = B.Block(ux, vx, Yma)
    ctx.set_sweep(9)
    got = B.sample_users_blocked(blk, sample_mt, alpha, mu, Lam, ctx=ctx, entity_tag=6)
    assert got.shape == (D, len(ux))
    z_t = ctx.zeros(len(ux), D)
    check(lib().bdf_normals(ctx.handle, 1, 6, 0, len(ux), D, _p(z_t)))
    ctx.sync()
    z = z_t.cpu().numpy().T
    MM = sample_mt[:, vx - 1]
    covar = np.linalg.inv(Lam + alpha * (MM @ MM.T))
    mean = covar @ (alpha * MM @ Yma + (Lam @ mu)[:, None])
    np.testing.assert_allclose(got, np.linalg.cholesky(covar) @ z + mean, rtol=1e-8, atol=1e-9)
    # the row kernel on the block as a dense relation (one factorisation per user) gives the same values
    rows = B.sample_users_blocked(blk, sample_mt, alpha, mu, Lam, ctx=ctx, entity_tag=6, shared=False)
    np.testing.assert_allclose(got, rows, rtol=1e-9, atol=1e-10)
    # larger blocks at every kernel width: shared covariance == per-row factorisation == the reference's expression
    for D2, M2, nv2, nu2 in ((5, 30, 7, 9), (32, 300, 150, 70), (64, 200, 90, 33)):
        smt = rng.standard_normal((D2, M2))
        vx2 = rng.choice(M2, nv2, replace=False) + 1
        Y2 = rng.standard_normal((nv2, nu2))
        A2 = rng.standard_normal((D2, D2))
        Lam2, mu2 = A2 @ A2.T / D2 + np.eye(D2), rng.standard_normal(D2)
        b2 = B.Block(np.arange(1, nu2 + 1), vx2, Y2)
        g2 = B.sample_users_blocked(b2, smt, 0.9, mu2, Lam2, ctx=ctx, entity_tag=3)
        r2 = B.sample_users_blocked(b2, smt, 0.9, mu2, Lam2, ctx=ctx, entity_tag=3, shared=False)
        np.testing.assert_allclose(g2, r2, rtol=1e-8, atol=1e-9)
        z2 = ctx.zeros(nu2, D2)
        check(lib().bdf_normals(ctx.handle, 1, 3, 0, nu2, D2, _p(z2)))
        ctx.sync()
        MM2 = smt[:, vx2 - 1]
        cov2 = np.linalg.inv(Lam2 + 0.9 * (MM2 @ MM2.T))
        np.testing.assert_allclose(g2, np.linalg.cholesky(cov2) @ z2.cpu().numpy().T + cov2 @ (0.9 * MM2 @ Y2 + (Lam2 @ mu2)[:, None]),
                                   rtol=1e-7, atol=1e-8)
    with pytest.raises(B.DimensionMismatch):
        B.Block(ux, vx, Yma.T)
    assert B.sample_users_blocked(B.Block([], vx, np.zeros((len(vx), 0))), sample_mt, alpha, mu, Lam, ctx=ctx).shape == (D, 0)


@pytest.mark.parametrize("D", [1, 3, 8, 10, 13, 16])
@pytest.mark.parametrize("coded", [True, False])
def test_four_rows_per_wave_equals_wave_per_row(B, ctx, D, coded):
    """D <= 16: the short rows of an entity of one two-mode relation sampled four to a wave (k_rows_small: 16 lanes and a
    column-per-lane system per row, bdf_ctx_set_small_rows) against the wave-per-row kernel on the same inputs -- the same
    sample up to the order of the floating-point sums; rows of 0 .. 48 observations take the new path (two chunks of 16
    included), 49 and more the old one in the same call; a row count that is no multiple of four; ratings (coded ids) and
    continuous values; with and without per-row prior means"""
    import ctypes as C
    from bdf_amd._lib import Term, check, lib
    rng = np.random.default_rng(100 + D)
    dims = [203, 90]
    deg = rng.integers(0, 40, dims[0])
    deg[:6] = [0, 1, 16, 17, 48, 49]
    deg[6:9] = [33, 150, 2]
    rows = np.repeat(np.arange(1, dims[0] + 1), deg)
    ids = np.stack([rows, rng.integers(1, dims[1] + 1, len(rows))], axis=1).astype(np.int64)
    vals = rng.integers(1, 6, len(rows)).astype(np.float64) if coded else rng.random(len(rows)) * 4 + 1
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    V = ctx.tensor(rng.standard_normal((dims[1], D)) * 0.5)
    A = rng.standard_normal((D, D))
    Lam = ctx.tensor(A @ A.T / max(D, 1) + np.eye(D))
    mu = ctx.tensor(rng.standard_normal(D))
    mu_rows = ctx.tensor(rng.standard_normal((dims[0], D)))
    p = lambda t: C.c_void_p(t.data_ptr())
    terms = (Term * 1)()
    terms[0].rel = dr.handle; terms[0].mode = 0; terms[0].alpha = 1.7; terms[0].mean_value = float(vals.mean())
    terms[0].factors[1] = V.data_ptr()
    for per_row in (False, True):
        outs = []
        for small in (0, 48):
            ctx.set_small_rows(small, 1)
            out = ctx.zeros(dims[0], D)
            ctx.set_sweep(5)
            check(lib().bdf_sample_rows(ctx.handle, D, dims[0], 1, terms, p(mu_rows if per_row else mu), 1 if per_row else 0, p(Lam), 3, 0, 1,
                                        p(out), None))
            ctx.sync()
            outs.append(out.cpu().numpy())
        assert np.isfinite(outs[1]).all() and np.abs(outs[1]).max() > 0.05
        np.testing.assert_allclose(outs[1], outs[0], rtol=1e-11, atol=1e-12)
        assert np.array_equal(outs[1][deg >= 49], outs[0][deg >= 49])          # the long rows: the same kernel either way
    ctx.set_small_rows(48, 8192)


@pytest.mark.parametrize("D", [1, 3, 8, 10, 12, 13, 16])
@pytest.mark.parametrize("coded", [True, False])
def test_four_rows_per_wave_against_oracle(B, O, ctx, D, coded):
    """k_rows_small (D <= 16, four rows per wave) against the CPU ORACLE (sample_user_basic, sampling.jl:200-212), not
    against the other kernel: rows of 0 / 1 / 15 / 16 / 17 / 32 / 48 observations on the new path (one, two and three chunks of
    16, ragged and full), 49 and more through k_rows in the same call; a row count that is no multiple of four; ratings
    (coded ids) and continuous values; shared and per-row prior means; both modes of the relation.  Row system 1e-12 through
    the dump, samples 1e-8 for the same normals."""
    from bdf_amd._lib import Term, check, lib
    rng = np.random.default_rng(300 + D)
    dims = [203, 90]
    deg = rng.integers(0, 49, dims[0])
    deg[:10] = [0, 1, 15, 16, 17, 32, 48, 49, 150, 2]
    rows = np.repeat(np.arange(1, dims[0] + 1), deg)
    ids = np.stack([rows, rng.integers(1, dims[1] + 1, len(rows))], axis=1).astype(np.int64)
    vals = rng.integers(1, 6, len(rows)).astype(np.float64) if coded else rng.random(len(rows)) * 4 + 1
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    facs = [rng.standard_normal((d, D)) * 0.5 for d in dims]
    ft = [ctx.tensor(f) for f in facs]
    A = rng.standard_normal((D, D))
    Lam = A @ A.T / max(D, 1) + np.eye(D)
    alpha, mean = 1.7, float(vals.mean())
    idx = O.index_build(ids, dims)
    ctx.set_small_rows(48, 1)
    try:
        for mode0 in (0, 1):
            N = dims[mode0]
            mu = rng.standard_normal(D)
            mu_rows = rng.standard_normal((N, D))
            terms = _dev_terms(B, ctx, [(dr, mode0, alpha, mean, [None if k == mode0 else ft[k] for k in (0, 1)], None)])
            ot = O.Term(ids, vals, dims, mode0, alpha, mean, [None if k == mode0 else facs[k] for k in (0, 1)], index=idx)
            Lam_t = ctx.tensor(Lam)
            for per_row in (False, True):
                m = mu_rows if per_row else mu
                mu_t = ctx.tensor(m)
                ctx.set_sweep(6)
                out_t = ctx.zeros(N, D)
                _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 4, out_t)
                exp = O.sample_rows(D, N, [ot], m, Lam, SEED, 6, 4)
                got = out_t.cpu().numpy()
                assert np.isfinite(got).all()
                np.testing.assert_allclose(got, exp, rtol=1e-8, atol=1e-9)
    finally:
        ctx.set_small_rows(48, 8192)
    dr.close()


@pytest.mark.parametrize("D", [17, 24, 30, 32, 33, 48, 64])
def test_lowrank_rows_against_oracle(B, O, ctx, D):
    """k_rows_lr4 (D > 16: rows of at most min(16, D / 2) observations by the low-rank sampler, four rows to a wave) against the oracle's statement of
    the same map (orc_sample_row_lowrank: D + n normals, n x n solve) on the same Philox normals, 1e-8; the longer rows of the
    same launch against the reference's map (they go through k_rows).  Rows of 0 .. 15 observations and more, a row count that
    is no multiple of four or sixteen, ratings and continuous values, both modes, shared and per-row prior means.  (That the low-rank map draws the reference's
    distribution is the deterministic CPU test test_lowrank_sampler_draws_the_reference_distribution.)"""
    rng = np.random.default_rng(500 + D)
    dims = [211, 90]
    deg = rng.integers(0, 18, dims[0])
    deg[:10] = [0, 1, 2, 4, 5, 14, 15, 16, 40, 250]
    deg[10:14] = [17, 8, 9, 12]
    deg[14:22] = [18, 24, 31, 32, 33, 20, 29, 64]       # D > 32: rows of 17 .. 32 observations by k_rows_lr32 (two observations per lane)
    rows = np.repeat(np.arange(1, dims[0] + 1), deg)
    ids = np.stack([rows, rng.integers(1, dims[1] + 1, len(rows))], axis=1).astype(np.int64)
    lr = min(32 if D > 32 else 16, D // 2)
    idx = O.index_build(ids, dims)
    Am = rng.standard_normal((D, D))
    Lam = Am @ Am.T / D + np.eye(D)
    mu = rng.standard_normal(D)
    for coded in (True, False):
        vals = rng.integers(1, 6, len(rows)).astype(np.float64) if coded else rng.random(len(rows)) * 4 + 1
        dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
        facs = [rng.standard_normal((d, D)) * 0.5 for d in dims]
        ft = [ctx.tensor(f) for f in facs]
        alpha, mean = 1.7, float(vals.mean())
        ctx.set_lowrank(-1, 0)
        try:
            for mode0 in (0, 1):
                N = dims[mode0]
                terms = _dev_terms(B, ctx, [(dr, mode0, alpha, mean, [None if k == mode0 else ft[k] for k in (0, 1)], None)])
                ot = O.Term(ids, vals, dims, mode0, alpha, mean, [None if k == mode0 else facs[k] for k in (0, 1)], index=idx)
                Lam_t, mu_t = ctx.tensor(Lam), ctx.tensor(mu)
                ctx.set_sweep(7)
                out_t = ctx.zeros(N, D)
                _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 5, out_t)
                got = out_t.cpu().numpy()
                exp = O.sample_rows_lowrank(D, N, [ot], mu, Lam, lr, SEED, 7, 5)
                assert np.isfinite(got).all()
                np.testing.assert_allclose(got, exp, rtol=1e-8, atol=1e-9)
                # per-row prior means (entity side information: mu + uhat_i, macau.jl:103-104): L' mu_i row by row
                mu_rows = rng.standard_normal((N, D))
                mur_t = ctx.tensor(mu_rows)
                out_r = ctx.zeros(N, D)
                _run_rows(B, ctx, D, N, terms, mur_t, Lam_t, 5, out_r)
                exp_r = O.sample_rows_lowrank(D, N, [ot], mu_rows, Lam, lr, SEED, 7, 5)
                np.testing.assert_allclose(out_r.cpu().numpy(), exp_r, rtol=1e-8, atol=1e-9)
                if mode0 == 0:
                    # switched off: the reference's map for every row
                    ctx.set_lowrank(0, 0)
                    out2 = ctx.zeros(N, D)
                    _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 5, out2)
                    ctx.set_lowrank(-1, 0)
                    lit = O.sample_rows(D, N, [ot], mu, Lam, SEED, 7, 5)
                    np.testing.assert_allclose(out2.cpu().numpy(), lit, rtol=1e-8, atol=1e-9)
                    cnt = np.bincount(ids[:, 0] - 1, minlength=N)
                    assert np.array_equal(out2.cpu().numpy()[cnt > lr], got[cnt > lr])       # the long rows: the same kernel either way
                    assert not np.allclose(lit[cnt <= lr], got[cnt <= lr])
        finally:
            ctx.set_lowrank(-1, 8192)
        dr.close()


@pytest.mark.parametrize("D,n", [(32, 0), (32, 3), (32, 16), (64, 10)])
def test_lowrank_row_moments(B, O, ctx, D, n):
    """>= 10^5 draws of one row by the low-rank sampler on the device: every row of a 512-row entity has the SAME observations
    (so the same conditional distribution) and its own random stream; 200 sweeps x 512 rows = 102,400 draws.  Sample mean within
    5 sigma / sqrt(draws) of inv(P) b in every coordinate; sample covariance within 3 % of inv(P) (relative Frobenius norm;
    the sampling error of a D x D covariance from 10^5 draws is ~ D / sqrt(draws) = 1 % at D = 32)."""
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(900 + D + n)
    N, M = 512, 40
    cols = rng.choice(M, size=max(n, 1), replace=False)[:n] + 1
    # (one more row, with 20 observations, so that the relation is never empty: it takes the reference's map and is not looked at)
    ids = np.concatenate([np.stack([np.repeat(np.arange(1, N + 1), n), np.tile(cols, N)], axis=1).reshape(-1, 2),
                          np.stack([np.full(20, N + 1), np.arange(1, 21)], axis=1)]).astype(np.int64)
    vals = np.concatenate([np.tile(rng.standard_normal(n), N), rng.standard_normal(20)])
    dims = [N + 1, M]
    V = rng.standard_normal((M, D)) * 0.6
    Am = rng.standard_normal((D, D))
    Lam = Am @ Am.T / D + np.eye(D)
    mu = rng.standard_normal(D) * 0.3
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    Vt = ctx.tensor(V)
    terms = _dev_terms(B, ctx, [(dr, 0, 2.0, 0.1, [None, Vt], None)])
    Lam_t, mu_t = ctx.tensor(Lam), ctx.tensor(mu)
    ctx.set_lowrank(16, 0)
    try:
        sweeps = 200
        import torch
        acc = torch.zeros(sweeps, N + 1, D, dtype=torch.float64, device=Lam_t.device)
        for s in range(sweeps):
            ctx.set_sweep(s + 1)
            out_t = acc[s]
            check(lib().bdf_sample_rows(ctx.handle, D, N + 1, 1, terms, _p(mu_t), 0, _p(Lam_t), 1, 0, 1, _p(out_t), None))
        ctx.sync()
    finally:
        ctx.set_lowrank(-1, 8192)
    draws = acc.cpu().numpy()[:, :N].reshape(-1, D)
    ot = O.Term(ids, vals, dims, 0, 2.0, 0.1, [None, V])
    P, b = O.row_system(D, [ot], 0, mu, Lam)
    cov = np.linalg.inv(P)
    mean = cov @ b
    nd = draws.shape[0]
    assert nd >= 100_000
    se = np.sqrt(np.diag(cov) / nd)
    assert np.all(np.abs(draws.mean(0) - mean) < 5 * se)
    emp = np.cov(draws.T)
    assert np.linalg.norm(emp - cov) / np.linalg.norm(cov) < 0.03
    # and it IS the low-rank map (not the reference's) that ran
    exp1 = O.sample_row_lowrank(D, [ot], 3, mu, Lam, O.lowrank_normals(SEED, 1, 1, 3, D, n))
    np.testing.assert_allclose(draws[3], exp1, rtol=1e-8, atol=1e-9)
    dr.close()


def _probe(tmp_path, env_extra, *args):
    """tests/env_probe.py in a fresh process with the environment given (switches the library reads once per process)"""
    import os, subprocess, sys
    out = str(tmp_path / "probe.npz")
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_probe.py"), args[0], out] +
                       [str(a) for a in args[1:]], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_lowrank_wave_per_row_kernel_with_a_padded_list(B, O, tmp_path):
    """BDF_LR_WAVE=1: the wave-per-row form of the low-rank sampler (k_rows_lr, rows of at most 15 observations) over a list that is
    padded to a multiple of four with row = -1 records -- which it must skip (round 5 ran it over them: stores before the factor
    matrix) -- against the oracle's low-rank function at 1e-8, shared and per-row prior means, a guard buffer untouched."""
    from env_probe import lowrank_problem
    D = 24
    lr = min(15, D // 2)
    for n_rows in range(203, 220):      # a list whose length is no multiple of four (the problem is a function of the seed and n_rows)
        ids, vals, facs, Lam, mu, mu_rows = lowrank_problem(515, [n_rows, 60], D)
        cnt = np.bincount(ids[:, 0] - 1, minlength=n_rows)
        if int((cnt <= lr).sum()) % 4 != 0:
            break
    dims = [n_rows, 60]
    got = _probe(tmp_path, {"BDF_LR_WAVE": "1"}, "lowrank", 515, n_rows, D)
    ot = O.Term(ids, vals, dims, 0, 1.7, float(vals.mean()), [None, facs[1]])
    assert int((cnt <= lr).sum()) % 4 != 0 and int(got["lr_rows"][0]) == 2 * int((cnt <= lr).sum())     # (two launches under the one tag add up)
    for name, m in (("shared", mu), ("per_row", mu_rows)):
        exp = O.sample_rows_lowrank(D, n_rows, [ot], m, Lam, lr, SEED, 7, 5)
        np.testing.assert_allclose(got[name], exp, rtol=1e-8, atol=1e-9)
        assert not got[name + "_guard"].any()


def test_normal_wishart_draw_moments(B, O, ctx):
    """Sampled moments of the device's Normal-Wishart draw (bdf_hyper_sample, the default map of the mean normals) over 12,000
    draws at D = 6 against rand(::NormalWishart)'s law (/root/reference/src/normal_wishart.jl:38-42):
    E[Lambda] = nu_N T_N with Var[Lambda_ij] = nu_N (T_ij^2 + T_ii T_jj), E[mu] = mu_N, Cov[mu] = inv(T_N) / (beta_N (nu_N - D - 1)).
    Means within 5 standard errors; the mean's covariance within 8 % (Frobenius)."""
    from bdf_amd._lib import check, lib
    D, N, n = 6, 40, 12000
    rng = np.random.default_rng(909)
    U = rng.standard_normal((N, D)) * 0.8 + rng.standard_normal(D) * 0.3
    mu0 = rng.standard_normal(D) * 0.1
    A = rng.standard_normal((D, D))
    Tinv = A @ A.T / D + np.eye(D)
    b0, nu = 2.0, D + 4.0
    S_t = ctx.tensor(U)
    sumU, UUt = ctx.zeros(D), ctx.zeros(D, D)
    check(lib().bdf_hyper_sums(ctx.handle, D, N, _p(S_t), None, _p(sumU), _p(UUt)))
    mu0_t, Tinv_t = ctx.tensor(mu0), ctx.tensor(Tinv)
    mus, Lams = ctx.zeros(n, D), ctx.zeros(n, D * D)
    for s in range(n):
        ctx.set_sweep(s + 1)
        check(lib().bdf_hyper_sample(ctx.handle, D, N, _p(sumU), _p(UUt), _p(mu0_t), b0, _p(Tinv_t), nu, 3,
                                     C.c_void_p(mus.data_ptr() + 8 * D * s), C.c_void_p(Lams.data_ptr() + 8 * D * D * s), None, None, None))
    ctx.sync()
    mu_N, beta_N, T_N, nu_N = O.hyper_params(U, mu0, b0, Tinv, nu)
    m = mus.cpu().numpy()
    L = Lams.cpu().numpy().reshape(n, D, D)
    assert np.isfinite(m).all() and np.isfinite(L).all()
    # one draw against the oracle on the same streams (the map), the rest by their moments (the law)
    mu_e, Lam_e = O.hyper_draw(mu_N, beta_N, T_N, nu_N, SEED, 17, 3)
    np.testing.assert_allclose(m[16], mu_e, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(L[16], Lam_e, rtol=1e-7, atol=1e-9)
    EL = nu_N * T_N
    seL = np.sqrt(nu_N * (T_N ** 2 + np.outer(np.diag(T_N), np.diag(T_N))) / n)
    assert np.all(np.abs(L.mean(0) - EL) < 5 * seL)
    cov_mu = np.linalg.inv(T_N) / (beta_N * (nu_N - D - 1))
    assert np.all(np.abs(m.mean(0) - mu_N) < 5 * np.sqrt(np.diag(cov_mu) / n))
    emp = np.cov(m.T)
    assert np.linalg.norm(emp - cov_mu) / np.linalg.norm(cov_mu) < 0.08
