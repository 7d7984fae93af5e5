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


# ---- BASELINE configurations C3, C4, C5 (M-C3, M-C4, M-C5) ------------------------------------------------
def c3_user_features(kind="iid", n_users=6040, n_feat=500):
    """M-C3 dense user side information 6040 x 500: i.i.d. N(0,1) (seed 4242, well conditioned: ~15-25 CG iterations), or
    the correlated variant Z W + 0.1 E with Z 6040 x 20, W 20 x 500 (seeds 4243 / 4244: F'F has 20 dominant directions)"""
    if kind == "iid":
        return np.random.default_rng(4242).standard_normal((n_users, n_feat))
    if kind != "correlated":
        raise ValueError("kind must be 'iid' or 'correlated'")
    r1, r2 = np.random.default_rng(4243), np.random.default_rng(4244)
    Z, W = r1.standard_normal((n_users, 20)), r1.standard_normal((20, n_feat))
    return Z @ W + 0.1 * r2.standard_normal((n_users, n_feat))


def c3_relation_data(B, kind="iid", ntest=500_000):
    """C3: Macau on MovieLens-1M with the dense user features of c3_user_features (BASELINE.json configs[2])"""
    rd, source = movielens_relation_data(B, ntest=ntest, seed=1, alpha=1.5, class_cut=2.5)
    rd.entities[0].F = c3_user_features(kind, n_users=rd.entities[0].count)
    return rd, source


def synth_ratings(n_rows, n_cols, nnz, seed=777, zipf_offset=100.0, test_fraction=0.01, k_begin=0):
    """M-C4 observations k_begin .. k_begin + nnz - 1 from the library's counter-based host generator (bdf_synth_ratings: rows
    uniform, columns Zipf-like p ~ 1 / (c + zipf_offset), ratings 1..5 from a planted rank-8 model).
    -> rows, cols (int32, 1-based), vals (float64), held (bool)"""
    import ctypes as C
    from . import _lib
    rows, cols = np.empty(nnz, dtype=np.int32), np.empty(nnz, dtype=np.int32)
    vals, held = np.empty(nnz, dtype=np.float64), np.zeros(nnz, dtype=np.uint8)
    _lib.check(_lib.lib().bdf_synth_ratings(C.c_uint64(seed), n_rows, n_cols, k_begin, k_begin + nnz, float(zipf_offset),
                                            float(test_fraction), rows.ctypes.data_as(_lib.c_i32p), cols.ctypes.data_as(_lib.c_i32p),
                                            vals.ctypes.data_as(_lib.c_dp), held.ctypes.data_as(C.c_void_p)))
    return rows, cols, vals, held.astype(bool)


def c4_relation_data(B, n_rows=10_000_000, n_cols=1_000_000, nnz=100_000_000, seed=777, alpha=2.0, zipf_offset=100.0,
                     test_fraction=0.01):
    """C4 (BASELINE.json configs[3]): synthetic n_rows x n_cols relation with nnz observations (1 % of them held out), BPMF,
    alpha = 2.  Int32 ids (FastIDF{Int32}, test/basic.jl:32): 100M observations are 0.8 GB of ids instead of 1.6."""
    rows, cols, vals, held = synth_ratings(n_rows, n_cols, nnz, seed, zipf_offset, test_fraction)
    keep = ~held
    ids = np.empty((int(keep.sum()), 2), dtype=np.int32, order="F")
    ids[:, 0], ids[:, 1] = rows[keep], cols[keep]
    users, items = B.Entity("users"), B.Entity("items")
    rel = B.Relation(B.IndexedDF((ids, vals[keep]), [n_rows, n_cols], names=["users", "items", "value"]), "ratings",
                     [users, items], class_cut=2.5)
    users.count, items.count = n_rows, n_cols
    if held.any():
        tid = np.stack([rows[held], cols[held]], axis=1).astype(np.int64)
        B.setTest(rel, (tid, vals[held]))
    B.setPrecision(rel, alpha)
    return B.RelationData(rel)


def c5_relation_data(B, nA=100_000, nB=64, nC=1_000, nT=500, n1=5_000_000, n2=1_000_000, n_feat=50_000, feat_per_row=50,
                     noise=0.1):
    """C5 (BASELINE.json configs[4], M-C5): entity A shared by a 3-mode relation A x B x C (planted rank-8 CP model, seed
    901) and a 2-mode relation A x T (seed 902); A has binary sparse features with feat_per_row ones per row (seed 903,
    SparseBinMatrixCSR layout); alpha 5 / 2; 1 % of the first relation held out.  Sizes scale down for the parity tests."""
    import scipy.sparse as sp
    R = 8
    rng = np.random.default_rng(901)
    fa, fb, fc = rng.standard_normal((nA, R)) * 0.7, rng.standard_normal((nB, R)) * 0.7, rng.standard_normal((nC, R)) * 0.7
    key = np.unique(rng.integers(0, nA * nB * nC, size=int(n1 * 1.02)))[:n1]
    ia, ib, ic = key // (nB * nC), (key // nC) % nB, key % nC
    v1 = np.sum(fa[ia] * fb[ib] * fc[ic], axis=1) + noise * rng.standard_normal(len(key))
    rng2 = np.random.default_rng(902)
    ft = rng2.standard_normal((nT, R)) * 0.7
    key2 = np.unique(rng2.integers(0, nA * nT, size=int(n2 * 1.02)))[:n2]
    ja, jt = key2 // nT, key2 % nT
    v2 = np.sum(fa[ja] * ft[jt], axis=1) + noise * rng2.standard_normal(len(key2))
    rng3 = np.random.default_rng(903)
    cols = rng3.integers(0, n_feat, size=(nA, feat_per_row))
    Fbin = sp.csr_matrix((np.ones(nA * feat_per_row), (np.repeat(np.arange(nA), feat_per_row), cols.ravel())), shape=(nA, n_feat))
    Fbin.data[:] = 1.0
    A = B.Entity("A", F=Fbin)
    Bn, Cn, Tn = B.Entity("B"), B.Entity("C"), B.Entity("T")
    r1 = B.Relation((np.stack([ia + 1, ib + 1, ic + 1], axis=1), v1), "abc", [A, Bn, Cn], dims=[nA, nB, nC])
    r2 = B.Relation((np.stack([ja + 1, jt + 1], axis=1), v2), "at", [A, Tn], dims=[nA, nT])
    B.assignToTest(r1, max(1, len(v1) // 100), rng=np.random.default_rng(5))
    B.setPrecision(r1, 5.0)
    B.setPrecision(r2, 2.0)
    rd = B.RelationData()
    B.addRelation(rd, r1)
    B.addRelation(rd, r2)
    assert len(A.relations) == 2            # note N1: the shared entity is conditioned on BOTH relations
    return rd, {"value_std": float(v1.std()), "noise": noise}
