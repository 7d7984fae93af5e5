"""GPU parity of the side-information path (K2-K5): Entity.F operators, AtA_mul_B!, cg_AtA / solve_cg2, solve_full,
sample_beta, sample_lambda_beta -- against the CPU oracle and against the dense algebra the reference's own tests use
(test/solver.jl, test/sparse_csr.jl, test/sparsebin_csr.jl, test/parallel_matrix.jl:41-109, test/heavy_copyto.jl:28-69).
Tolerances: products 1e-12 relative; CG solutions 1e-9 relative (the solver stops at eps*numF like the reference);
sampled beta 1e-7 relative for the same normals.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 1234


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _colmajor(ctx, A):
    """numpy (rows, cols) -> torch tensor holding the column-major matrix (shape (cols, rows))"""
    return ctx.tensor(np.ascontiguousarray(np.asarray(A, dtype=np.float64).T))


def _from_colmajor(t):
    return t.cpu().numpy().T


def _reference_pattern():
    """rows = [1:200; 151:350], cols = [151:350; 1:2:399] -- test/sparsebin_csr.jl:4-5, test/parallel_matrix.jl:41-42"""
    rows = np.concatenate([np.arange(1, 201), np.arange(151, 351)]).astype(np.int32)
    cols = np.concatenate([np.arange(151, 351), np.arange(1, 400, 2)]).astype(np.int32)
    A = np.zeros((350, 399))
    np.add.at(A, (rows - 1, cols - 1), 1.0)
    return rows, cols, A


def _operators(B, ctx, rng):
    import scipy.sparse as sp
    rows, cols, Abin = _reference_pattern()
    Fd = rng.standard_normal((120, 37))
    S = sp.random(90, 64, density=0.15, random_state=3, format="coo")
    return {
        "dense": (B.FeatOperator(ctx, Fd), Fd),
        "csr": (B.FeatOperator(ctx, S), S.toarray()),
        "bincoo": (B.FeatOperator(ctx, B.SparseBinMatrix(rows, cols)), Abin),
        "bincsr": (B.FeatOperator(ctx, B.SparseBinMatrixCSR(rows, cols)), Abin),
    }


def test_feature_products_all_kinds(B, ctx):
    rng = np.random.default_rng(0)
    for name, (op, A) in _operators(B, ctx, rng).items():
        m, n = A.shape
        assert (op.m, op.n) == (m, n)
        X = rng.standard_normal((n, 5))
        Y = rng.standard_normal((m, 3))
        np.testing.assert_allclose(_from_colmajor(op.mul(_colmajor(ctx, X))), A @ X, rtol=1e-12, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(_from_colmajor(op.mul(_colmajor(ctx, Y), transpose=True)), A.T @ Y, rtol=1e-12, atol=1e-12,
                                   err_msg=name)
        # AtA_mul_B! (parallel_cg.jl:7-14), lambda = 0.1 as in test/parallel_matrix.jl:72-92
        got = _from_colmajor(op.AtA_mul(_colmajor(ctx, X), 0.1))
        np.testing.assert_allclose(got, A.T @ (A @ X) + 0.1 * X, rtol=1e-12, atol=1e-11, err_msg=name)
        op.close()


def test_sparse_csr_literal(B, ctx):
    """test/sparse_csr.jl:28-34"""
    F = B.sparse_csr([1, 2, 2, 4], [2, 1, 3, 3], [0.1, 0.2, 0.15, 0.3])
    op = B.FeatOperator(ctx, F)
    z = np.array([0.3, -1.2, 0.7])
    dense = np.zeros((4, 3))
    dense[[0, 1, 1, 3], [1, 0, 2, 2]] = [0.1, 0.2, 0.15, 0.3]
    np.testing.assert_allclose(_from_colmajor(op.mul(_colmajor(ctx, z[:, None])))[:, 0], dense @ z, rtol=1e-14)
    op.close()


def _sample_beta(B, ctx, op, D, sample, mu, Lam, lb, use_ff, tol, sample_lambda=False, nu=1e-3, mu_h=1.0, tag=3, maxiter=0):
    from bdf_amd._lib import check, lib
    import torch
    S_t, mu_t, Lam_t = ctx.tensor(sample), ctx.tensor(mu), ctx.tensor(Lam)
    lb_t = ctx.tensor([lb])
    beta_t, rhs_t = ctx.zeros(D, op.n), ctx.zeros(D, op.n)
    it_t = torch.zeros(D, dtype=torch.int32, device=ctx.device)
    check(lib().bdf_sample_beta(ctx.handle, op.handle, D, _p(S_t), _p(mu_t), _p(Lam_t), _p(lb_t), int(use_ff),
                                float("nan") if tol is None else tol, maxiter, int(sample_lambda), nu, mu_h, tag,
                                _p(beta_t), _p(rhs_t), _p(it_t)))
    ctx.sync()
    return _from_colmajor(beta_t), _from_colmajor(rhs_t), it_t.cpu().numpy(), float(lb_t.item())


@pytest.mark.parametrize("kind,D", [("dense", 8), ("csr", 5), ("bincoo", 3), ("dense", 32)])
def test_sample_beta_cg_and_direct_match_oracle(B, O, ctx, kind, D):
    rng = np.random.default_rng(11)
    op, A = _operators(B, ctx, rng)[kind]
    N, numF = A.shape
    sample = rng.standard_normal((N, D))
    mu = rng.standard_normal(D) * 0.1
    M = rng.standard_normal((D, D))
    Lam = M @ M.T / D + np.eye(D)
    lb = 0.75
    ctx.set_sweep(6)
    if kind == "dense":
        ofeat = O.Feat.from_dense(A)
    else:
        r, c = np.nonzero(A)
        ofeat = O.Feat.from_csr(r, c, A[r, c], N, numF)
    beta_e, rhs_e, it_e = O.sample_beta(ofeat, sample, mu, Lam, lb, False, None, SEED, 6, 3)
    beta, rhs, iters, _ = _sample_beta(B, ctx, op, D, sample, mu, Lam, lb, False, None)
    np.testing.assert_allclose(rhs, rhs_e, rtol=1e-9, atol=1e-9)
    scale = np.abs(beta_e).max()
    np.testing.assert_allclose(beta, beta_e, rtol=1e-7, atol=1e-9 * scale)
    # both solve (F'F + lb I) beta = rhs to the reference's tolerance: compare with the direct solve as test/heavy_copyto.jl does
    direct = np.linalg.solve(A.T @ A + lb * np.eye(numF), rhs_e)
    np.testing.assert_allclose(beta, direct, rtol=1e-7, atol=1e-9 * scale)
    assert np.all(np.abs(iters - it_e) <= 2), (iters, it_e)
    # FF path (solve_full): same system, direct
    beta_ff, _, _, _ = _sample_beta(B, ctx, op, D, sample, mu, Lam, lb, True, None)
    np.testing.assert_allclose(beta_ff, direct, rtol=1e-7, atol=1e-9 * scale)
    op.close()


@pytest.mark.parametrize("kind,D", [("csr", 32), ("bin", 6), ("csr", 17)])
def test_sample_beta_cg_long_columns_row_major_state(B, O, ctx, kind, D):
    """Sparse features with more than 2,048 columns and no F'F: the conjugate-gradient state lives ROW-major from the solve's first
    launch to its last (k_cg_rm_*: the sparse products take and leave it as it is -- no transposes around them) -- 3,000 x 2,600
    features, 9 entries per row, real-valued (CSR) and binary, D = 32 (sixteen 16-byte lanes per gathered row), 6 and 17 (an odd
    row length: the general sparse kernel): rhs and beta against the oracle's literal cg_AtA (src/parallel_cg.jl:63-94) on the same
    noise streams, beta against the direct solve of the same system, the per-column iteration counts within one."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2600 + D)
    N, numF, per = 3000, 2600, 9
    rows = np.repeat(np.arange(N), per)
    cols = np.concatenate([np.sort(rng.choice(numF, per, replace=False)) for _ in range(N)])
    vals = np.ones(len(rows)) if kind == "bin" else rng.standard_normal(len(rows))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(N, numF))
    op = B.FeatOperator(ctx, B.SparseBinMatrix(N, numF, rows + 1, cols + 1) if kind == "bin" else B.sparse_csr(rows + 1, cols + 1, vals, N, numF))
    sample = rng.standard_normal((N, D))
    mu = rng.standard_normal(D) * 0.1
    M = rng.standard_normal((D, D))
    Lam = M @ M.T / D + np.eye(D)
    lb = 2.5
    ctx.set_sweep(8)
    ofeat = O.Feat.from_csr(rows, cols, vals, N, numF)
    beta_e, rhs_e, it_e = O.sample_beta(ofeat, sample, mu, Lam, lb, False, 1e-10, SEED, 8, 3)
    beta, rhs, iters, _ = _sample_beta(B, ctx, op, D, sample, mu, Lam, lb, False, 1e-10)
    np.testing.assert_allclose(rhs, rhs_e, rtol=1e-9, atol=1e-9)
    scale = np.abs(beta_e).max()
    np.testing.assert_allclose(beta, beta_e, rtol=1e-7, atol=1e-8 * scale)
    direct = np.linalg.solve((A.T @ A).toarray() + lb * np.eye(numF), rhs_e)
    np.testing.assert_allclose(beta, direct, rtol=1e-6, atol=1e-7 * scale)
    assert np.all(np.abs(iters - it_e) <= 1), (iters, it_e)
    op.close()


@pytest.mark.parametrize("kind", ["dense", "csr"])
def test_cg_out_of_iterations_is_reported_not_raised(B, ctx, kind):
    """cg_AtA (parallel_cg.jl:73-93) returns a column that ran out of iterations as it stands; the library does the same and
    leaves BDF_WARN_CG_MAXITER for the host (Context.sync turns it into a RuntimeWarning); a solve that converges leaves none"""
    import warnings
    rng = np.random.default_rng(12)
    op, A = _operators(B, ctx, rng)[kind]
    N, numF = A.shape
    D = 4
    sample, mu, Lam = rng.standard_normal((N, D)), np.zeros(D), np.eye(D)
    ctx.set_sweep(2)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, _, it_ok, _ = _sample_beta(B, ctx, op, D, sample, mu, Lam, 0.5, False, None)       # converges well before numF iterations
    assert it_ok.max() < numF
    with pytest.warns(RuntimeWarning, match="conjugate-gradient"):
        beta, _, it_cut, _ = _sample_beta(B, ctx, op, D, sample, mu, Lam, 0.5, False, None, maxiter=2)
    assert np.all(it_cut == 2) and np.all(np.isfinite(beta))
    with warnings.catch_warnings():                       # the condition was taken, not sticky
        warnings.simplefilter("error")
        ctx.sync()
    op.close()


def test_cg_reference_cases(B, O, ctx):
    """cg_AtA against the direct solve: lambda 0.5 (test/parallel_matrix.jl:107-109), 0.75 with tol 1e-6 and several
    right-hand sides (test/heavy_copyto.jl:28-50), driven through sample_beta with the noise switched off by a huge Lambda"""
    rows, cols, A = _reference_pattern()
    op = B.FeatOperator(ctx, B.SparseBinMatrix(rows, cols))
    rng = np.random.default_rng(2)
    D = 3
    N, numF = A.shape
    target = rng.random((N, D))
    Lam = np.eye(D) * 1e30                    # noise ~ 1e-15: rhs == F' target
    for lb, tol in ((0.5, None), (0.75, 1e-6)):
        beta, rhs, iters, _ = _sample_beta(B, ctx, op, D, target, np.zeros(D), Lam, lb, False, tol)
        np.testing.assert_allclose(rhs, A.T @ target, rtol=1e-9, atol=1e-9)
        exact = np.linalg.solve(A.T @ A + lb * np.eye(numF), rhs)
        np.testing.assert_allclose(beta, exact, rtol=1e-5 if tol else 1e-8, atol=1e-6 if tol else 1e-9)
        assert np.all(iters > 0) and np.all(iters <= numF)
    op.close()


def test_solve_full_literal_shape(B, ctx):
    """test/solver.jl:4-12: X 1000 x 50, y 50 x 3, lambda 0.75 -- through the FF path of sample_beta"""
    rng = np.random.default_rng(5)
    X = rng.random((1000, 50))
    op = B.FeatOperator(ctx, X)
    D = 3
    target = rng.random((1000, D))
    beta, rhs, _, _ = _sample_beta(B, ctx, op, D, target, np.zeros(D), np.eye(D) * 1e30, 0.75, True, None)
    exact = np.linalg.solve(X.T @ X + 0.75 * np.eye(50), X.T @ target)
    np.testing.assert_allclose(beta, exact, rtol=1e-8, atol=1e-10)
    op.close()


def test_lambda_beta_draw_matches_oracle(B, O, ctx):
    rng = np.random.default_rng(9)
    D, N, numF = 6, 80, 12
    F = rng.standard_normal((N, numF))
    op = B.FeatOperator(ctx, F)
    sample = rng.standard_normal((N, D))
    mu = np.zeros(D)
    M = rng.standard_normal((D, D))
    Lam = M @ M.T / D + np.eye(D)
    ctx.set_sweep(4)
    beta, _, _, lb_new = _sample_beta(B, ctx, op, D, sample, mu, Lam, 1.5, True, None, sample_lambda=True, nu=1e-3, mu_h=1.0, tag=7)
    exp = O.sample_lambda_beta(beta, Lam, 1e-3, 1.0, SEED, 4, 7)
    assert lb_new > 0
    np.testing.assert_allclose(lb_new, exp, rtol=1e-8)
    op.close()


def test_uhat_and_feature_hyper_terms(B, ctx):
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(13)
    D, N, numF = 10, 55, 7
    F = rng.standard_normal((N, numF))
    op = B.FeatOperator(ctx, F)
    beta = rng.standard_normal((numF, D))
    mu = rng.standard_normal(D)
    beta_t, mu_t = _colmajor(ctx, beta), ctx.tensor(mu)
    uhat_t, mm_t = ctx.zeros(N, D), ctx.zeros(N, D)
    check(lib().bdf_uhat(ctx.handle, op.handle, D, _p(beta_t), _p(mu_t), _p(uhat_t), _p(mm_t)))
    WI = np.eye(D) * 2.0
    WI_t, lb_t, T_t = ctx.tensor(WI), ctx.tensor([0.6]), ctx.zeros(D, D)
    check(lib().bdf_hyper_feature_terms(ctx.handle, D, numF, _p(beta_t), _p(WI_t), _p(lb_t), _p(T_t)))
    ctx.sync()
    np.testing.assert_allclose(uhat_t.cpu().numpy(), F @ beta, rtol=1e-12, atol=1e-12)        # (F beta)' as D x N
    np.testing.assert_allclose(mm_t.cpu().numpy(), F @ beta + mu, rtol=1e-12, atol=1e-12)      # mu .+ uhat
    np.testing.assert_allclose(T_t.cpu().numpy(), WI + beta.T @ beta * 0.6, rtol=1e-12)
    op.close()


def test_sample_alpha_and_sse_match_oracle(B, O, ctx):
    """sample_alpha (sampling.jl:129-134) on the device: err' err over the training table, then the Wishart(1-d) draw"""
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(21)
    D, dims, n = 6, [30, 20], 400
    ids = np.stack([rng.integers(1, d + 1, n) for d in dims], axis=1)
    y = rng.standard_normal(n)
    facs = [rng.standard_normal((d, D)) * 0.4 for d in dims]
    ft = [ctx.tensor(f) for f in facs]
    pairs = B.DevicePairs(ctx, ids, y)
    mean = 0.3
    stats = pairs.sse(D, ft, mean)
    ctx.sync()
    pred = O.predict(ids, facs, mean)
    sse = float(np.sum((pred - y) ** 2))
    np.testing.assert_allclose(stats.cpu().numpy()[1], sse, rtol=1e-12)
    lin = rng.standard_normal(n)
    lin_t = ctx.tensor(lin)
    stats = pairs.sse(D, ft, mean, lin_t)
    ctx.sync()
    np.testing.assert_allclose(stats.cpu().numpy()[1], np.sum((pred - mean + lin - y) ** 2), rtol=1e-12)
    ctx.set_sweep(9)
    alpha_t = ctx.zeros(1)
    stats = pairs.sse(D, ft, mean)
    check(lib().bdf_sample_alpha(ctx.handle, 1.0, 2.0, n, C.c_void_p(stats.data_ptr() + 8), 1, _p(alpha_t)))
    ctx.sync()
    np.testing.assert_allclose(alpha_t.item(), O.sample_alpha(1.0, 2.0, n, sse, SEED, 9, 1), rtol=1e-9)
    pairs.close()


@pytest.mark.parametrize("numF", [2, 40, 90])
def test_sample_beta_rel_matches_oracle(B, O, ctx, numF):
    """sample_beta_rel (sampling.jl:322-337) + linear_values (macau.jl:91)"""
    from bdf_amd._lib import check, lib
    rng = np.random.default_rng(31 + numF)
    D, dims, n = 5, [25, 18], 500
    ids = np.stack([rng.integers(1, d + 1, n) for d in dims], axis=1)
    y = rng.standard_normal(n) * 2
    facs = [rng.standard_normal((d, D)) * 0.5 for d in dims]
    ft = [ctx.tensor(f) for f in facs]
    Fm = rng.standard_normal((n, numF))
    op = B.FeatOperator(ctx, Fm)
    pairs = B.DevicePairs(ctx, ids, y)
    mean, alpha, lam = 0.2, 1.7, 0.8
    beta_t, lin_t, rhs_t = ctx.zeros(numF), ctx.zeros(n), ctx.zeros(numF)
    fp = (C.c_void_p * 2)(*[f.data_ptr() for f in ft])
    ctx.set_sweep(4)
    check(lib().bdf_sample_beta_rel(ctx.handle, op.handle, pairs.handle, D, fp, mean, alpha, lam, 2, _p(beta_t), _p(lin_t), _p(rhs_t)))
    ctx.sync()
    res = y - O.predict(ids, facs, mean)
    beta_e, rhs_e = O.sample_beta_rel(O.Feat.from_dense(Fm), res, alpha, lam, SEED, 4, 2)
    np.testing.assert_allclose(rhs_t.cpu().numpy(), rhs_e, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(beta_t.cpu().numpy(), beta_e, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(lin_t.cpu().numpy(), mean + Fm @ beta_t.cpu().numpy(), rtol=1e-11, atol=1e-11)
    op.close()
    pairs.close()


@pytest.mark.parametrize("N,numF,D", [(700, 200, 12), (900, 500, 32), (600, 333, 17), (600, 500, 1)])
def test_cg_one_launch_solve_matches_the_two_launch_solve_and_the_oracle(B, O, N, numF, D):
    """k_cg_resident (the whole batched conjugate-gradient solve in ONE launch for a resident F'F of at most 512 features and at
    most min(32, ceil(numF / 16)) columns: row slices of the operator in registers, two grid-wide hand-overs per iteration)
    against the two-launches-per-iteration solve (BDF_CG_RESIDENT=0, read once per process: a child each) -- the same
    per-column iteration counts (cg_AtA's stopping rule, src/parallel_cg.jl:63-94) and beta to 1e-9 -- and against the oracle's
    literal cg_AtA on the same noise streams; a loose tolerance makes the columns stop at different iterations."""
    import os, subprocess, sys, tempfile, textwrap
    code = textwrap.dedent('''
        import numpy as np, sys, ctypes as C, torch
        sys.path.insert(0, %r)
        import bdf_amd as B
        from bdf_amd._lib import check, lib
        N, numF, D = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        rng = np.random.default_rng(N + numF + D)
        F = rng.standard_normal((N, numF)) * (0.2 + rng.random(numF))
        sample = rng.standard_normal((N, D)); mu = rng.standard_normal(D) * 0.1
        M = rng.standard_normal((D, D)); Lam = M @ M.T / D + np.eye(D)
        ctx = B.Context(seed=77); ctx.set_sweep(4)
        op = B.FeatOperator(ctx, F)
        p = lambda t: C.c_void_p(t.data_ptr())
        out = {}
        for name, tol in (("tight", float("nan")), ("loose", 1e-4)):
            S_t, mu_t, Lam_t, lb_t = ctx.tensor(sample), ctx.tensor(mu), ctx.tensor(Lam), ctx.tensor([0.6])
            beta_t, rhs_t = ctx.zeros(D, numF), ctx.zeros(D, numF)
            it_t = torch.zeros(D, dtype=torch.int32, device=ctx.device)
            check(lib().bdf_sample_beta(ctx.handle, op.handle, D, p(S_t), p(mu_t), p(Lam_t), p(lb_t), 0, tol, 0, 0, 1e-3, 1.0, 3,
                                        p(beta_t), p(rhs_t), p(it_t)))
            try:
                ctx.sync()
            except RuntimeWarning:
                pass
            out["beta_" + name] = beta_t.cpu().numpy(); out["rhs_" + name] = rhs_t.cpu().numpy(); out["it_" + name] = it_t.cpu().numpy()
        np.savez(sys.argv[1], F=F, sample=sample, mu=mu, Lam=Lam, **out)
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for resident in ("1", "0"):
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "o.npz")
            env = dict(os.environ, BDF_CG_RESIDENT=resident)
            subprocess.run([sys.executable, "-W", "ignore", "-c", code, f, str(N), str(numF), str(D)], check=True, env=env, timeout=600)
            got[resident] = dict(np.load(f))
    a, b = got["1"], got["0"]
    for name in ("tight", "loose"):
        assert np.array_equal(a["it_" + name], b["it_" + name]), (name, a["it_" + name], b["it_" + name])
        np.testing.assert_array_equal(a["rhs_" + name], b["rhs_" + name])
        np.testing.assert_allclose(a["beta_" + name], b["beta_" + name], rtol=1e-9, atol=1e-11)
    assert a["it_loose"].max() < a["it_tight"].max()
    # the oracle's literal cg_AtA on the same right-hand side (two products with F per iteration): same solution to the solver's tolerance
    beta_o, _, _ = O.sample_beta(O.Feat.from_dense(a["F"]), a["sample"], a["mu"], a["Lam"], 0.6, False, None, 77, 4, 3)
    np.testing.assert_allclose(a["beta_tight"].reshape(D, numF).T, beta_o, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("kind", ["bin", "csr"])
def test_sparse_products_in_column_panels(B, ctx, kind):
    """A gathered operand of 8 MiB and more is taken in column panels (k_feat.hip, spmm: 12,288 rows of it per launch, the rows'
    running sums carried through the output) -- 40,000 x 40,000 with 24 entries per row, 32 columns: four panels in both
    directions.  Against scipy at 1e-12 (sparsebin_csr.jl:49-63 / sparse_csr.jl semantics); and the same matrix given with every
    row's entries in a shuffled order -- which the library takes in ONE pass: the panels must keep the order of a row's sum, so an
    operand whose rows are not in column order is not panelled -- gives the same values to rounding."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    m = n = 40_000
    per = 24
    rows = np.repeat(np.arange(m), per)
    cols = np.concatenate([np.sort(rng.choice(n, per, replace=False)) for _ in range(m)])
    vals = np.ones(len(rows)) if kind == "bin" else rng.standard_normal(len(rows))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
    def make(r, c, v):
        if kind == "bin":
            return B.FeatOperator(ctx, B.SparseBinMatrix(m, n, r + 1, c + 1))
        return B.FeatOperator(ctx, B.sparse_csr(r + 1, c + 1, v, m, n))
    op = make(rows, cols, vals)
    X = rng.standard_normal((n, 32))
    Y = rng.standard_normal((m, 32))
    fwd = _from_colmajor(op.mul(_colmajor(ctx, X)))
    tr = _from_colmajor(op.mul(_colmajor(ctx, Y), transpose=True))
    np.testing.assert_allclose(fwd, A @ X, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(tr, A.T @ Y, rtol=1e-12, atol=1e-12)
    # the same matrix, every row's entries in a shuffled order: one pass (no panels), the same values to rounding
    perm = np.concatenate([r0 + rng.permutation(per) for r0 in range(0, m * per, per)])
    op2 = make(rows[perm], cols[perm], vals[perm])
    fwd2 = _from_colmajor(op2.mul(_colmajor(ctx, X)))
    np.testing.assert_allclose(fwd2, fwd, rtol=1e-12, atol=1e-12)
    op.close(); op2.close()
