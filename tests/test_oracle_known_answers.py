"""The CPU oracle (oracle/bdf_oracle.c) pinned on every known-answer the reference's own tests hold for this path, on the
Random123 Philox vectors, and on independent numpy fp64 algebra.  (The reference stores no seeded sampled values, so the
random STREAM is unpinned -- see the oracle header; the maps from normals to samples are checked here deterministically.)
"""
import numpy as np
import pytest


def test_indexeddf_literals(O):
    """test/basic.jl:7-19: A=[2,2,3], B=[1,3,4], dims [4,4]"""
    ids = np.array([[2, 1], [2, 3], [3, 4]])
    (rp1, ri1), (rp2, ri2) = O.index_build(ids, [4, 4])
    assert rp1.tolist() == [0, 0, 2, 3, 3] and ri1.tolist() == [1, 2, 3]      # entity 2 of mode 1 -> rows 1,2 ; entity 3 -> row 3
    assert rp2.tolist() == [0, 1, 1, 2, 3] and ri2.tolist() == [1, 2, 3]
    with pytest.raises(IndexError):
        O.index_build(np.array([[5, 1]]), [4, 4])


def test_index_is_stable_in_table_order(O):
    rng = np.random.default_rng(0)
    dims = [9, 7, 4]
    ids = np.stack([rng.integers(1, d + 1, 500) for d in dims], axis=1)
    for m, (rp, ri) in enumerate(O.index_build(ids, dims)):
        for j in range(dims[m]):
            expect = np.nonzero(ids[:, m] == j + 1)[0] + 1          # push!(index[mode][j], i) in row order
            assert np.array_equal(ri[rp[j]:rp[j + 1]], expect)


def test_rep_int_literal(O):
    """test/basic.jl:20"""
    assert O.rep_int([2, 4, 1, 10], [3, 2, 1, 0]).tolist() == [2, 2, 2, 4, 4, 1]


def test_philox_random123_vectors(O):
    """Random123 kat_vectors, philox4x32-10"""
    assert [hex(x) for x in O.philox4x32_10([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_normals_and_gamma_moments(O):
    z = np.concatenate([O.normals(7, 3, 1, 2, r, 32) for r in range(3000)])
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and abs((z ** 4).mean() - 3) < 0.15
    for a in (0.5, 3.5, 40.0):
        g = np.array([O.gamma(5, 1, 0, i, a) for i in range(20000)])
        assert abs(g.mean() - a) < 0.05 * max(a, 1) and abs(g.var() - a) < 0.1 * max(a, 1)


def test_row_sample_is_the_reference_map(O):
    """sample_user_basic (sampling.jl:200-212): covar = inv(Lambda + alpha MM MM'); mu = covar (alpha MM rr + Lambda mu_u);
    x = chol(covar)' z + mu -- against numpy; the tensor form (:215-234) and the sum over relations (:266-289) too"""
    rng = np.random.default_rng(1)
    D = 7
    dimsA, dimsB = [11, 8, 5], [11, 9]
    idsA = np.stack([rng.integers(1, d + 1, 200) for d in dimsA], axis=1)
    idsB = np.stack([rng.integers(1, d + 1, 80) for d in dimsB], axis=1)
    vA, vB = rng.standard_normal(200), rng.standard_normal(80)
    F = [rng.standard_normal((d, D)) for d in dimsA]
    FB = rng.standard_normal((9, D))
    linB = rng.standard_normal(80)
    M = rng.standard_normal((D, D))
    Lam = M @ M.T + np.eye(D)
    mu = rng.standard_normal(D)
    tA = O.Term(idsA, vA, dimsA, 0, 1.5, 0.2, [None, F[1], F[2]])
    tB = O.Term(idsB, vB, dimsB, 0, 0.5, -0.1, [None, FB], linear_values=linB)
    for row in range(11):
        z = rng.standard_normal(D)
        P = Lam.copy()
        b = Lam @ mu
        sel = idsA[:, 0] == row + 1
        MM = (F[1][idsA[sel, 1] - 1] * F[2][idsA[sel, 2] - 1]).T
        P += 1.5 * MM @ MM.T
        b += 1.5 * MM @ (vA[sel] - 0.2)
        sel = idsB[:, 0] == row + 1
        MM = FB[idsB[sel, 1] - 1].T
        P += 0.5 * MM @ MM.T
        b += 0.5 * MM @ (vB[sel] - linB[sel])
        cov = np.linalg.inv(P)
        expect = np.linalg.cholesky(cov) @ z + cov @ b
        x, m = O.sample_row(D, [tA, tB], row, mu, Lam, z)
        np.testing.assert_allclose(x, expect, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(m, cov @ b, rtol=1e-10, atol=1e-12)
        P2, b2 = O.row_system(D, [tA, tB], row, mu, Lam)
        np.testing.assert_allclose(P2, P, rtol=1e-12)
        np.testing.assert_allclose(b2, b, rtol=1e-12, atol=1e-12)


def test_conditional_normal_wishart_parameters_and_draw(O):
    """ConditionalNormalWishart (sampling.jl:116-127) against numpy; rand(::NormalWishart) (normal_wishart.jl:38-42):
    Lambda = (L_T A)(L_T A)' with the Bartlett A built from the oracle's own streams, mu = mu_N + chol(inv(Lam)/kappa)' z"""
    rng = np.random.default_rng(2)
    D, N = 5, 40
    U = rng.standard_normal((N, D)) + 0.3
    mu0 = rng.standard_normal(D) * 0.1
    M = rng.standard_normal((D, D))
    Tinv = M @ M.T + np.eye(D)
    b0, nu = 2.0, 7.0
    mu_N, beta_N, T_N, nu_N = O.hyper_params(U, mu0, b0, Tinv, nu)
    NU, NS = U.sum(0), U.T @ U
    assert nu_N == nu + N and beta_N == b0 + N
    mu_e = (b0 * mu0 + NU) / (b0 + N)
    np.testing.assert_allclose(mu_N, mu_e, rtol=1e-13)
    T_e = np.linalg.inv(Tinv + NS + b0 * np.outer(mu0, mu0) - beta_N * np.outer(mu_e, mu_e))
    np.testing.assert_allclose(T_N, T_e, rtol=1e-10, atol=1e-14)
    mu, Lam = O.hyper_draw(mu_N, beta_N, T_N, nu_N, 99, 4, 2, mean_map="reference")
    A = np.zeros((D, D))
    for i in range(D):
        A[i, :i] = O.normals(99, 4, O.P_NW_NORMAL, 2, i, D)[:i]
        A[i, i] = np.sqrt(2.0 * O.gamma(99, 4, 2, i, 0.5 * (nu_N - i)))
    Z = np.linalg.cholesky((T_N + T_N.T) / 2) @ A
    np.testing.assert_allclose(Lam, Z @ Z.T, rtol=1e-9, atol=1e-12)
    z = O.normals(99, 4, O.P_NW_MEAN, 2, 0, D)
    np.testing.assert_allclose(mu, mu_N + np.linalg.cholesky(np.linalg.inv(Lam) / beta_N) @ z, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("D", [1, 3, 10, 32, 64])
def test_hyper_mean_through_the_wishart_factor_draws_the_reference_distribution(O, D):
    """The library's default map from the D mean normals to mu (round 5; oracle: orc_hyper_draw2 with mean_map 1) is NOT the
    reference's function of z -- rand(::NormalWishart) draws mu = mu_N + chol(inv(Lam) / beta_N)' z (normal_wishart.jl:40) -- but it
    draws the SAME conditional distribution, deterministically checkable because both maps are affine in z: mu = mu_N + S z with
    S S' = inv(beta_N Lam).  Here: (i) the oracle's two maps return the same Lambda (the mean map does not touch the Wishart draw);
    (ii) each equals mu_N + S z for its own S, recomputed in numpy from the Bartlett matrix the streams give; (iii) both S S' equal
    inv(beta_N Lam) to 1e-10 relative; (iv) the new S is Z^-T / sqrt(beta_N) with Z = chol(T_N)' A lower triangular, Lam = Z Z'."""
    rng = np.random.default_rng(500 + D)
    N = 50 + 3 * D
    U = rng.standard_normal((N, D)) * 0.7 + 0.2
    mu0 = rng.standard_normal(D) * 0.1
    M = rng.standard_normal((D, D))
    Tinv = M @ M.T / D + np.eye(D)
    mu_N, beta_N, T_N, nu_N = O.hyper_params(U, mu0, 2.0, Tinv, float(D))
    seed, sweep, tag = 4321, 17, 3
    mu_ref, Lam_ref = O.hyper_draw(mu_N, beta_N, T_N, nu_N, seed, sweep, tag, mean_map="reference")
    mu_fac, Lam_fac = O.hyper_draw(mu_N, beta_N, T_N, nu_N, seed, sweep, tag, mean_map="factor")
    assert np.array_equal(Lam_ref, Lam_fac)
    A = np.zeros((D, D))
    for i in range(D):
        A[i, :i] = O.normals(seed, sweep, O.P_NW_NORMAL, tag, i, D)[:i]
        A[i, i] = np.sqrt(2.0 * O.gamma(seed, sweep, tag, i, 0.5 * (nu_N - i)))
    Z = np.linalg.cholesky((T_N + T_N.T) / 2) @ A
    assert np.allclose(Z, np.tril(Z)) and (np.diag(Z) > 0).all()
    np.testing.assert_allclose(Lam_fac, Z @ Z.T, rtol=1e-9, atol=1e-12)
    z = O.normals(seed, sweep, O.P_NW_MEAN, tag, 0, D)
    cov = np.linalg.inv(beta_N * Lam_fac)
    S_ref = np.linalg.cholesky((cov + cov.T) / 2)
    S_fac = np.linalg.inv(Z).T / np.sqrt(beta_N)
    scale = np.abs(cov).max()
    np.testing.assert_allclose(S_ref @ S_ref.T, cov, rtol=0, atol=1e-10 * scale)
    np.testing.assert_allclose(S_fac @ S_fac.T, cov, rtol=0, atol=1e-10 * scale)
    np.testing.assert_allclose(mu_ref, mu_N + S_ref @ z, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(mu_fac, mu_N + S_fac @ z, rtol=1e-8, atol=1e-10)
    if D > 1:
        assert not np.allclose(mu_ref, mu_fac)          # two different functions of z


def _reference_pattern():
    rows = np.concatenate([np.arange(1, 201), np.arange(151, 351)])
    cols = np.concatenate([np.arange(151, 351), np.arange(1, 400, 2)])
    A = np.zeros((350, 399))
    np.add.at(A, (rows - 1, cols - 1), 1.0)
    return rows, cols, A


def test_binary_operators_reference_pattern(O):
    """test/sparsebin_csr.jl:4-23, test/parallel_matrix.jl:41-92: A_mul_B!, At_mul_B, AtA_mul_B! (lambda 0.1) vs sparse(rows,cols,1.0)"""
    rows, cols, A = _reference_pattern()
    rng = np.random.default_rng(3)
    x, y = rng.random(399), rng.random(350)
    for f in (O.Feat.from_bincsr(rows - 1, cols - 1, 350, 399), O.Feat.from_bincoo(rows - 1, cols - 1, 350, 399)):
        np.testing.assert_allclose(f.mul(x), A @ x, rtol=1e-13)
        np.testing.assert_allclose(f.tmul(y), A.T @ y, rtol=1e-13)
        np.testing.assert_allclose(f.AtA_mul_B(x, 0.1), A.T @ (A @ x) + 0.1 * x, rtol=1e-12)


def test_sparse_csr_literal_and_random(O):
    """test/sparse_csr.jl:4-34"""
    f = O.Feat.from_csr(np.array([1, 2, 2, 4]) - 1, np.array([2, 1, 3, 3]) - 1, [0.1, 0.2, 0.15, 0.3], 4, 3)
    z = np.array([0.4, 0.9, -0.3])
    dense = np.zeros((4, 3))
    dense[[0, 1, 1, 3], [1, 0, 2, 2]] = [0.1, 0.2, 0.15, 0.3]
    np.testing.assert_allclose(f.mul(z), dense @ z, rtol=1e-14)
    import scipy.sparse as sp
    X = sp.random(50, 100, 0.1, random_state=1, format="coo")
    f = O.Feat.from_csr(X.row, X.col, X.data, 50, 100)
    rng = np.random.default_rng(4)
    y1, y2 = rng.random(100), rng.random(50)
    np.testing.assert_allclose(f.mul(y1), X @ y1, rtol=1e-13)
    np.testing.assert_allclose(f.tmul(y2), X.T @ y2, rtol=1e-13)


def test_cg_and_solve_full_against_direct(O):
    """test/parallel_matrix.jl:107-109 (lambda 0.5), test/heavy_copyto.jl:28-50 (0.75, tol 1e-6), test/solver.jl:4-20"""
    rows, cols, A = _reference_pattern()
    f = O.Feat.from_bincoo(rows - 1, cols - 1, 350, 399)
    rng = np.random.default_rng(5)
    x = rng.random(399)
    AA = A.T @ A
    beta, it = f.cg_AtA(x, 0.5)
    np.testing.assert_allclose(beta, np.linalg.solve(AA + 0.5 * np.eye(399), x), rtol=1e-9, atol=1e-11)
    assert 0 < it <= 399
    beta2, _ = f.cg_AtA(x, 0.75, tol=1e-6)
    np.testing.assert_allclose(beta2, np.linalg.solve(AA + 0.75 * np.eye(399), x), rtol=1e-4, atol=1e-6)
    Xd = rng.random((1000, 50))
    y = rng.random((50, 3))
    np.testing.assert_allclose(O.solve_full(Xd.T @ Xd, y, 0.75), np.linalg.solve(Xd.T @ Xd + 0.75 * np.eye(50), y), rtol=1e-9)
    Ad = rng.random((500, 20))
    xd = rng.random(20)
    np.testing.assert_allclose(O.Feat.from_dense(Ad).AtA_mul_B(xd, 0.5), (Ad.T @ Ad + 0.5 * np.eye(20)) @ xd, rtol=1e-12)


def test_sample_beta_structure(O):
    """sample_beta (sampling.jl:291-312): rhs = F'((U - mu)' + E1) + sqrt(lb) E2 with E rows = chol(inv(Lambda))' z;
    CG (per column, tol eps*numF) and solve_full agree with the direct solve"""
    rng = np.random.default_rng(6)
    N, numF, D = 70, 9, 4
    F = rng.standard_normal((N, numF))
    S = rng.standard_normal((N, D))
    mu = rng.standard_normal(D) * 0.1
    M = rng.standard_normal((D, D))
    Lam = M @ M.T + np.eye(D)
    Lc = np.linalg.cholesky(np.linalg.inv(Lam))
    E1 = np.stack([Lc @ O.normals(3, 2, O.P_BETA_E1, 5, i, D) for i in range(N)])
    E2 = np.stack([Lc @ O.normals(3, 2, O.P_BETA_E2, 5, i, D) for i in range(numF)])
    rhs_e = F.T @ (S - mu + E1) + np.sqrt(0.8) * E2
    beta_cg, rhs, iters = O.sample_beta(O.Feat.from_dense(F), S, mu, Lam, 0.8, False, None, 3, 2, 5)
    np.testing.assert_allclose(rhs, rhs_e, rtol=1e-9, atol=1e-10)
    direct = np.linalg.solve(F.T @ F + 0.8 * np.eye(numF), rhs_e)
    np.testing.assert_allclose(beta_cg, direct, rtol=1e-8, atol=1e-10)
    assert np.all(iters <= numF)
    beta_ff, _, _ = O.sample_beta(O.Feat.from_dense(F), S, mu, Lam, 0.8, True, None, 3, 2, 5)
    np.testing.assert_allclose(beta_ff, direct, rtol=1e-9, atol=1e-11)


def test_predict_identity(O):
    """pred = sum_k prod_modes sample + mean (sampling.jl:9-45; test/basic.jl:112, test/tensor.jl:26-31)"""
    rng = np.random.default_rng(7)
    dims, D = [6, 5, 3], 4
    facs = [rng.standard_normal((d, D)) for d in dims]
    ids = np.array([[4, 2, 1], [1, 5, 3]])
    p = O.predict(ids, facs, 0.7)
    for q, (i, j, k) in enumerate(ids):
        assert np.isclose(p[q], np.sum(facs[0][i - 1] * facs[1][j - 1] * facs[2][k - 1]) + 0.7)


def test_oracle_sample_beta_rel_is_the_reference_formula():
    """sample_beta_rel (src/sampling.jl:322-337) restated with the oracle's own normals: the C routine equals
    (alpha F'F + lambda I) \\ (alpha F'(res + alpha^-1/2 z1) + sqrt(lambda) z2) in numpy"""
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    n, numF = 120, 7
    Fm = rng.standard_normal((n, numF))
    res = rng.standard_normal(n)
    alpha, lam, seed, sweep, tag = 2.5, 0.6, 99, 3, 4
    beta, rhs = O.sample_beta_rel(O.Feat.from_dense(Fm), res, alpha, lam, seed, sweep, tag)
    z1 = np.array([O.normals(seed, sweep, O.P_BETA_REL1, 0x800000 | tag, i, 1)[0] for i in range(n)])
    z2 = np.array([O.normals(seed, sweep, O.P_BETA_REL2, 0x800000 | tag, f, 1)[0] for f in range(numF)])
    rhs_np = alpha * Fm.T @ (res + z1 / np.sqrt(alpha)) + np.sqrt(lam) * z2
    np.testing.assert_allclose(rhs, rhs_np, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(beta, np.linalg.solve(alpha * Fm.T @ Fm + lam * np.eye(numF), rhs_np), rtol=1e-10)


@pytest.mark.parametrize("D", [4, 10, 30, 32, 64])
def test_lowrank_sampler_draws_the_reference_distribution(O, D):
    """The second row sampler (bdf_oracle.c orc_sample_row_lowrank, followed by the HIP library's k_rows_lr for rows of few
    observations) is NOT the reference's map from normals to the sample (sampling.jl:207-211) -- it must draw the same
    conditional distribution N(inv(P_i) b_i, inv(P_i)).  The map is affine in its D + n normals, x = m + S z, so this is a
    deterministic statement: m == inv(P_i) b_i and S S' == inv(P_i), checked to 1e-10 for every n in 0 .. D/2 + 1 (and a
    two-relation row), with shared and per-row prior means irrelevant to S."""
    rng = np.random.default_rng(D)
    n_rows = D // 2 + 2
    dims = [n_rows, 50]
    deg = np.arange(n_rows)                       # row r has r observations: 0 .. D/2 + 1
    rows = np.repeat(np.arange(1, n_rows + 1), deg)
    ids = np.stack([rows, rng.integers(1, 51, len(rows))], axis=1)
    vals = rng.standard_normal(len(rows))
    V = rng.standard_normal((50, D))
    A = rng.standard_normal((D, D))
    Lam = A @ A.T / D + np.eye(D)
    mu = rng.standard_normal(D)
    t = O.Term(ids, vals, dims, 0, 1.7, 0.2, [None, V])
    ids2 = np.stack([rng.integers(1, n_rows + 1, 3 * n_rows), rng.integers(1, 8, 3 * n_rows)], axis=1)
    t2 = O.Term(ids2, rng.standard_normal(3 * n_rows), [n_rows, 7], 0, 0.6, -0.1, [None, rng.standard_normal((7, D))],
                linear_values=rng.standard_normal(3 * n_rows))
    for terms in ([t], [t, t2]):
        for row in range(n_rows):
            P, b = O.row_system(D, terms, row, mu, Lam)
            m, S = O.lowrank_map(D, terms, row, mu, Lam)
            assert S.shape == (D, D + O.row_count(terms, row))
            cov = np.linalg.inv(P)
            np.testing.assert_allclose(m, cov @ b, rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(S @ S.T, cov, rtol=1e-10, atol=1e-10)
    # and the dispatch the library uses: rows above the threshold take the reference's map, bit for bit
    full = O.sample_rows(D, n_rows, [t], mu, Lam, 7, 3, 2)
    mixed = O.sample_rows_lowrank(D, n_rows, [t], mu, Lam, 3, 7, 3, 2)
    assert np.array_equal(mixed[4:], full[4:]) and not np.allclose(mixed[:4], full[:4])
