"""Workload builders shared by bench.py and the tests (host-side data plumbing, no numerics of the hot path).

MovieLens: the reference's bundled data/movielens_1m.mat (copied to data/ in this repo; MATLAB v5 file holding the
sparse rating matrix X 6040 x 3952 and the binary side-information matrices Fu, Fv -- docs/index.md:34-60).
The held-out split is the portable rule of BASELINE.md section 4: number the COO entries k = 0.. in the reference's
order (column-major findnz, RelationData.jl:165-171) and hold out the `ntest` entries with the smallest
splitmix64(k + 0x9E3779B97F4A7C15 * seed).
"""
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOVIELENS_PATH = os.path.join(_ROOT, "data", "movielens_1m.mat")
_G = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + _G
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def split_test_ids(nnz, ntest, seed=1):
    """1-based row numbers of the held-out entries (sorted ascending)"""
    with np.errstate(over="ignore"):
        h = splitmix64(np.arange(nnz, dtype=np.uint64) + _G * np.uint64(seed))
    order = np.argsort(h, kind="stable")
    return np.sort(order[:ntest]).astype(np.int64) + 1


def load_movielens(path=MOVIELENS_PATH):
    """-> dict with X (scipy CSC), Fu, Fv (scipy CSR)"""
    import scipy.io
    d = scipy.io.loadmat(path)
    return {"X": d["X"].tocsc(), "Fu": d["Fu"].tocsr(), "Fv": d["Fv"].tocsr()}


def synthetic_movielens_like(seed=0, n_users=6040, n_movies=3952, nnz=1000209):
    """Same shape and density as MovieLens-1M when the data file is absent: ratings 1..5 from a planted rank-8 model."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    pop = 1.0 / (np.arange(n_movies) + 50.0)
    cols = rng.choice(n_movies, size=int(nnz * 1.15), p=pop / pop.sum())
    rows = rng.integers(0, n_users, size=len(cols))
    key = np.unique(rows.astype(np.int64) * n_movies + cols)[:nnz]
    rows, cols = key // n_movies, key % n_movies
    U, V = rng.standard_normal((n_users, 8)) * 0.5, rng.standard_normal((n_movies, 8)) * 0.5
    vals = np.clip(np.round(3.5 + np.sum(U[rows] * V[cols], axis=1) + 0.5 * rng.standard_normal(len(rows))), 1, 5)
    return {"X": sp.csc_matrix((vals, (rows, cols)), shape=(n_users, n_movies)), "Fu": None, "Fv": None}


def replicate_users(X, test_ids, replicas):
    """The weak-scaling workload of bench.py --gpus N: the rating matrix stacked `replicas` times along the users (disjoint
    user blocks rating the same movies) and the held-out entries of every block.  X: CSC; test_ids: 1-based entry numbers
    in X's column-major order.  Entry k of X (column c, position p of the n_c entries of c) is entry
    replicas*colptr[c] + r*n_c + p of the stacked matrix for block r."""
    import scipy.sparse as sp
    X = X.tocsc()
    big = sp.vstack([X] * replicas).tocsc()
    colptr = X.indptr.astype(np.int64)
    k = np.asarray(test_ids, dtype=np.int64) - 1
    c = np.searchsorted(colptr, k, side="right") - 1
    n_c, p = colptr[c + 1] - colptr[c], k - colptr[c]
    ids = np.concatenate([replicas * colptr[c] + r * n_c + p for r in range(replicas)])
    return big, np.sort(ids) + 1


def movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5, with_features=False, path=MOVIELENS_PATH,
                            replicas=1):
    """The docs' MovieLens set-up (docs/index.md:42-56): entities users/movies, relation ratings, held-out test set,
    precision alpha.  Returns (RelationData, source) with source 'movielens_1m.mat' or 'synthetic'.  replicas > 1: the
    users (ratings, held-out entries, features) repeated in that many disjoint blocks (replicate_users)."""
    if os.path.exists(path):
        d, source = load_movielens(path), "movielens_1m.mat"
    else:
        d, source = synthetic_movielens_like(), "synthetic"
    X, test_ids = d["X"], (split_test_ids(d["X"].nnz, ntest, seed) if ntest else None)
    Fu = d["Fu"]
    if replicas > 1:
        import scipy.sparse as sp
        X, test_ids = replicate_users(X, test_ids if ntest else np.zeros(0, dtype=np.int64), replicas)
        Fu = sp.vstack([Fu] * replicas).tocsr() if Fu is not None else None
        source += f" x{replicas} user blocks"
    users = B.Entity("users", F=Fu if with_features else None)
    movies = B.Entity("movies", F=d["Fv"] if with_features else None)
    ratings = B.Relation(X, "ratings", [users, movies], class_cut=class_cut)
    if ntest:
        assert B.numData(ratings) == X.nnz
        B.assignToTest(ratings, test_ids)
    B.setPrecision(ratings, alpha)
    return B.RelationData(ratings), source
