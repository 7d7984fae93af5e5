"""Generates tests/golden/*.npz -- the golden vectors SURVEY.md 8(c) asks the build to make for itself.

   python tests/golden/make_golden.py        (needs numpy only, plus the C oracle for the vectors marked [oracle])

rows_small.npz   a tiny 3-mode + 2-mode problem sharing entity 0; per row of entity 0 the system the reference forms in
                 sample_user2 (src/sampling.jl:266-289): P_i = Lambda + sum_r alpha_r MM MM', b_i = Lambda mu + sum_r alpha_r
                 MM (values - baseline), the conditional mean P_i^-1 b_i and the sample chol(inv(P_i))' z + mean for a
                 stored z -- all from an INDEPENDENT numpy fp64 computation (inv + cholesky, the reference's own steps).
nw_small.npz     ConditionalNormalWishart parameters (src/sampling.jl:116-127) for a stored U, numpy.
index_basic.npz  the IndexedDF literal of the reference's test/basic.jl:7-19 (A=[2,2,3], B=[1,3,4], dims [4,4]): per-mode
                 row pointers and row lists, written out by hand from the test's asserted lists.
philox.npz       [oracle] Philox4x32-10 draws and the normals derived from them for fixed (seed, sweep, purpose, entity,
                 row): the stream contract between CPU oracle and device (the Random123 known-answer vectors that pin the
                 generator itself are literals in tests/test_oracle_known_answers.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def rows_small():
    rng = np.random.default_rng(20240)
    D = 6
    dimsA, dimsB = [9, 7, 4], [9, 5]
    idsA = np.stack([rng.integers(1, d + 1, 120) for d in dimsA], axis=1).astype(np.int64)
    idsB = np.stack([rng.integers(1, d + 1, 40) for d in dimsB], axis=1).astype(np.int64)
    idsA[idsA[:, 0] == 4, 0] = 3                       # row 4 (1-based) of entity 0 has no 3-mode observation
    vA, vB = rng.standard_normal(120), rng.standard_normal(40)
    FA = [rng.standard_normal((d, D)) for d in dimsA]
    FB1 = rng.standard_normal((dimsB[1], D))
    M = rng.standard_normal((D, D))
    Lam = M @ M.T + np.eye(D)
    mu = rng.standard_normal(D)
    alphaA, meanA, alphaB, meanB = 1.5, 0.2, 0.5, -0.1
    z = rng.standard_normal((dimsA[0], D))
    P = np.zeros((dimsA[0], D, D)); b = np.zeros((dimsA[0], D)); mean = np.zeros_like(b); x = np.zeros_like(b)
    for row in range(dimsA[0]):
        Pi, bi = Lam.copy(), Lam @ mu
        sel = idsA[:, 0] == row + 1
        MM = (FA[1][idsA[sel, 1] - 1] * FA[2][idsA[sel, 2] - 1]).T            # sampling.jl:277-280
        Pi += alphaA * MM @ MM.T
        bi += alphaA * MM @ (vA[sel] - meanA)
        sel = idsB[:, 0] == row + 1
        MM = FB1[idsB[sel, 1] - 1].T
        Pi += alphaB * MM @ MM.T
        bi += alphaB * MM @ (vB[sel] - meanB)
        cov = np.linalg.inv(Pi)                                                 # sampling.jl:284
        P[row], b[row], mean[row] = Pi, bi, cov @ bi
        x[row] = np.linalg.cholesky(cov) @ z[row] + mean[row]                   # sampling.jl:288
    np.savez(os.path.join(HERE, "rows_small.npz"), D=D, dimsA=dimsA, dimsB=dimsB, idsA=idsA, idsB=idsB, vA=vA, vB=vB,
             FA0=FA[0], FA1=FA[1], FA2=FA[2], FB1=FB1, Lambda=Lam, mu=mu, alphaA=alphaA, meanA=meanA, alphaB=alphaB,
             meanB=meanB, z=z, P=P, b=b, mean=mean, x=x)


def nw_small():
    rng = np.random.default_rng(20241)
    D, N = 5, 40
    U = rng.standard_normal((N, D)) * 0.8 + rng.standard_normal(D)
    mu0 = rng.standard_normal(D) * 0.1
    A = rng.standard_normal((D, D))
    Tinv = A @ A.T / D + np.eye(D)
    b0, nu = 2.0, D + 3.0
    Ubar, S = U.mean(0), U.T @ U
    beta_N, nu_N = b0 + N, nu + N
    mu_N = (b0 * mu0 + U.sum(0)) / beta_N                                       # sampling.jl:120-122
    W = Tinv + S + b0 * np.outer(mu0, mu0) - beta_N * np.outer(mu_N, mu_N)      # sampling.jl:124
    np.savez(os.path.join(HERE, "nw_small.npz"), U=U, mu0=mu0, Tinv=Tinv, b0=b0, nu=nu, mu_N=mu_N, beta_N=beta_N, nu_N=nu_N,
             T_N_inv=W, T_N=np.linalg.inv(W), Ubar=Ubar)


def index_basic():
    # test/basic.jl:7-19: X = IndexedDF(DataFrame(A=[2,2,3], B=[1,3,4], C=[0.,-1.,0.5]), [4,4])
    # getData(X,1,2) -> rows 1,2 ; getData(X,1,3) -> row 3 ; getData(X,2,1) -> row 1 ; (2,3) -> row 2 ; (2,4) -> row 3
    np.savez(os.path.join(HERE, "index_basic.npz"),
             ids=np.array([[2, 1], [2, 3], [3, 4]], dtype=np.int64), values=np.array([0.0, -1.0, 0.5]), dims=np.array([4, 4]),
             rowptr0=np.array([0, 0, 2, 3, 3], dtype=np.int64), rowids0=np.array([1, 2, 3], dtype=np.int64),
             rowptr1=np.array([0, 1, 1, 2, 3], dtype=np.int64), rowids1=np.array([1, 2, 3], dtype=np.int64))


def philox():
    from oracle import oracle as O
    cases = [(1234, 5, 1, 9, 0), (1234, 5, 1, 9, 12345678901), (7, 0, 4, 3, 31), (2 ** 40 + 3, 77, 7, 0xABCDEF, 0)]
    draws = np.array([[O.draw(s, sw, p, e, r, pair) for pair in range(4)] for (s, sw, p, e, r) in cases], dtype=np.uint32)
    normals = np.array([O.normals(s, sw, p, e, r, 8) for (s, sw, p, e, r) in cases])
    np.savez(os.path.join(HERE, "philox.npz"), cases=np.array(cases, dtype=np.uint64), draws=draws, normals=normals)


if __name__ == "__main__":
    rows_small(); nw_small(); index_basic(); philox()
    print("written:", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
