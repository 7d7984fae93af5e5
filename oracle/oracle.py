"""ctypes front-end of the CPU oracle (oracle/bdf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of bdf_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.

All matrices follow the reference's layout: column-major, an entity's sample is D x N
(numpy: array of shape (N, D), C-contiguous == D x N column-major).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)

P_ROW, P_BETA_E1, P_BETA_E2, P_NW_NORMAL, P_GAMMA_N, P_GAMMA_U, P_NW_MEAN, P_BETA_REL1, P_BETA_REL2 = 1, 2, 3, 4, 5, 6, 7, 8, 9


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_gamma.restype = C.c_double
        _LIB.orc_sample_lambda_beta.restype = C.c_double
        _LIB.orc_sample_alpha.restype = C.c_double
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OrcTerm(C.Structure):
    _fields_ = [("n_modes", C.c_int), ("mode", C.c_int), ("nnz", C.c_int64),
                ("ids", c_i64p), ("values", c_dp), ("rowptr", c_i64p), ("rowids", c_i64p),
                ("alpha", C.c_double), ("mean_value", C.c_double), ("linear_values", c_dp),
                ("factors", C.POINTER(c_dp))]


class OrcFeat(C.Structure):
    _fields_ = [("kind", C.c_int), ("m", C.c_int64), ("n", C.c_int64), ("nnz", C.c_int64),
                ("dense", c_dp), ("rowptr", c_i64p), ("colind", c_i32p), ("vals", c_dp),
                ("rows", c_i32p), ("cols", c_i32p)]


# ---------------------------------------------------------------------------------------
def index_build(ids, dims):
    """IndexedDF index (IndexedDF.jl:10-21). ids: (nnz, n_modes) 1-based. Returns per mode
    (rowptr[N+1], rowids[nnz]) with rowids the 1-based COO row numbers in original order."""
    ids = np.asfortranarray(np.asarray(ids, dtype=np.int64))
    nnz, n_modes = ids.shape if ids.ndim == 2 else (0, len(dims))
    dims = np.asarray(dims, dtype=np.int64)
    rps = [np.zeros(int(d) + 1, dtype=np.int64) for d in dims]
    ris = [np.zeros(max(nnz, 1), dtype=np.int64) for _ in dims]
    rp_arr = (c_i64p * n_modes)(*[r.ctypes.data_as(c_i64p) for r in rps])
    ri_arr = (c_i64p * n_modes)(*[r.ctypes.data_as(c_i64p) for r in ris])
    rc = lib().orc_index_build(n_modes, dims.ctypes.data_as(c_i64p), C.c_int64(nnz),
                               ids.ctypes.data_as(c_i64p), rp_arr, ri_arr)
    if rc != 0:
        raise IndexError("id outside 1..dims (BoundsError in the reference)")
    return [(rps[m], ris[m][:nnz]) for m in range(n_modes)]


def rep_int(x, times):
    x = np.asarray(x, dtype=np.int64)
    times = np.asarray(times, dtype=np.int64)
    out = np.zeros(int(times.sum()), dtype=np.int64)
    lib().orc_rep_int(x.ctypes.data_as(c_i64p), times.ctypes.data_as(c_i64p), C.c_int64(len(x)),
                      out.ctypes.data_as(c_i64p))
    return out


def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(ctr.ctypes.data_as(c_u32p), key.ctypes.data_as(c_u32p), out.ctypes.data_as(c_u32p))
    return out


def draw(seed, sweep, purpose, entity, row, pair):
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_draw(C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(purpose), C.c_uint32(entity),
                   C.c_uint64(row), C.c_uint32(pair), out.ctypes.data_as(c_u32p))
    return out


def normals(seed, sweep, purpose, entity, row, n):
    z = np.zeros(n + 1, dtype=np.float64)
    lib().orc_normals(C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(purpose), C.c_uint32(entity),
                      C.c_uint64(row), C.c_int(n), _dp(z))
    return z[:n]


def gamma(seed, sweep, entity, g, a):
    return lib().orc_gamma(C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(entity), C.c_uint64(g), C.c_double(a))


# ---------------------------------------------------------------------------------------
class Term:
    """One relation's view of the entity being sampled (keeps numpy buffers alive)."""

    def __init__(self, ids, values, dims, mode, alpha, mean_value, factors, linear_values=None, index=None):
        self.ids = np.asfortranarray(np.asarray(ids, dtype=np.int64))
        self.values = _f64(values)
        self.nnz, self.n_modes = self.ids.shape
        self.mode = mode
        idx = index if index is not None else index_build(self.ids, dims)
        self.rowptr, self.rowids = idx[mode]
        self.rowids = np.ascontiguousarray(self.rowids if len(self.rowids) else np.zeros(1, dtype=np.int64))
        self.factors = [None if f is None else _f64(f) for f in factors]
        self.lin = None if linear_values is None else _f64(linear_values)
        self._fp = (c_dp * self.n_modes)(*[None if f is None else _dp(f) for f in self.factors])
        self.alpha, self.mean_value = float(alpha), float(mean_value)

    def struct(self):
        return OrcTerm(self.n_modes, self.mode, self.nnz, self.ids.ctypes.data_as(c_i64p), _dp(self.values),
                       self.rowptr.ctypes.data_as(c_i64p), self.rowids.ctypes.data_as(c_i64p),
                       self.alpha, self.mean_value, None if self.lin is None else _dp(self.lin), self._fp)


def _terms(terms):
    return (OrcTerm * len(terms))(*[t.struct() for t in terms])


def row_system(D, terms, row, mu_i, Lambda):
    P = np.zeros((D, D), dtype=np.float64)
    b = np.zeros(D, dtype=np.float64)
    mu_i, Lambda = _f64(mu_i), _f64(Lambda)
    lib().orc_row_system(D, len(terms), _terms(terms), C.c_int64(row), _dp(mu_i), _dp(Lambda), _dp(P), _dp(b))
    return P.T.copy(), b          # P is symmetric; returned as numpy (i,j)


def sample_row(D, terms, row, mu_i, Lambda, z):
    x = np.zeros(D)
    m = np.zeros(D)
    mu_i, Lambda, z = _f64(mu_i), _f64(Lambda), _f64(z)
    rc = lib().orc_sample_row(D, len(terms), _terms(terms), C.c_int64(row), _dp(mu_i), _dp(Lambda), _dp(z), _dp(x), _dp(m))
    if rc:
        raise np.linalg.LinAlgError("row system not positive definite")
    return x, m


def sample_rows(D, N, terms, mu, Lambda, seed, sweep, entity_tag, out=None, row_begin=0, row_end=None, nthreads=1):
    """All rows [row_begin,row_end) of an entity (sampling.jl:181-198 / 251-264). Returns (N, D)."""
    mu, Lambda = _f64(mu), _f64(Lambda)
    if out is None:
        out = np.zeros((N, D), dtype=np.float64)
    row_end = N if row_end is None else row_end
    rc = lib().orc_sample_rows(D, C.c_int64(row_begin), C.c_int64(row_end), len(terms), _terms(terms), _dp(mu),
                               int(mu.ndim == 2), _dp(Lambda), C.c_uint64(seed), C.c_uint32(sweep),
                               C.c_uint32(entity_tag), _dp(out), nthreads)
    if rc:
        raise np.linalg.LinAlgError("row system not positive definite")
    return out


def row_count(terms, row):
    return int(sum(t.rowptr[row + 1] - t.rowptr[row] for t in terms))


def sample_row_lowrank(D, terms, row, mu_i, Lambda, z):
    """The second ("low-rank") row sampler (bdf_oracle.c, orc_sample_row_lowrank): z holds D + n normals."""
    x = np.zeros(D)
    mu_i, Lambda, z = _f64(mu_i), _f64(Lambda), _f64(z)
    assert len(z) >= D + row_count(terms, row)
    rc = lib().orc_sample_row_lowrank(D, len(terms), _terms(terms), C.c_int64(row), _dp(mu_i), _dp(Lambda), _dp(z), _dp(x))
    if rc:
        raise np.linalg.LinAlgError("row system not positive definite" if rc == -1 else "too many observations for the low-rank sampler")
    return x


def lowrank_normals(seed, sweep, entity_tag, row, D, n):
    """the D + n normals of the low-rank sampler for a row, in the order sample_row_lowrank takes them (orc_lowrank_normals)"""
    z = np.zeros(D + n + 1)
    scratch = np.zeros(2 * (D + n) + 70)
    lib().orc_lowrank_normals(C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(entity_tag), C.c_uint64(row), int(D), int(n), _dp(z), _dp(scratch))
    return z[:D + n]


def lowrank_map(D, terms, row, mu_i, Lambda):
    """dump hook: the low-rank sampler is affine in its normals, x = m + S z; returns (m, S) with S of shape (D, D + n)"""
    n = row_count(terms, row)
    m = sample_row_lowrank(D, terms, row, mu_i, Lambda, np.zeros(D + n))
    S = np.zeros((D, D + n))
    for k in range(D + n):
        e = np.zeros(D + n)
        e[k] = 1.0
        S[:, k] = sample_row_lowrank(D, terms, row, mu_i, Lambda, e) - m
    return m, S


def sample_rows_lowrank(D, N, terms, mu, Lambda, lr_max, seed, sweep, entity_tag, out=None, row_begin=0, row_end=None, nthreads=1):
    """All rows as the HIP library samples them with the low-rank sampler on: rows of at most lr_max observations by the
    low-rank sampler (normals 0 .. D+n-1 of the row's stream), the others by the reference's map. Returns (N, D)."""
    mu, Lambda = _f64(mu), _f64(Lambda)
    if out is None:
        out = np.zeros((N, D), dtype=np.float64)
    row_end = N if row_end is None else row_end
    rc = lib().orc_sample_rows_lowrank(D, C.c_int64(row_begin), C.c_int64(row_end), len(terms), _terms(terms), _dp(mu),
                                       int(mu.ndim == 2), _dp(Lambda), int(lr_max), C.c_uint64(seed), C.c_uint32(sweep),
                                       C.c_uint32(entity_tag), _dp(out), nthreads)
    if rc:
        raise np.linalg.LinAlgError("row system not positive definite")
    return out


# ---------------------------------------------------------------------------------------
def hyper_params(U, mu0, b0, Tinv, nu):
    """ConditionalNormalWishart (sampling.jl:116-127). U: (N, D). -> mu_N, beta_N, T_N, nu_N"""
    U = _f64(U)
    N, D = U.shape
    mu0, Tinv = _f64(mu0), _f64(Tinv)
    mu_N = np.zeros(D)
    T_N = np.zeros((D, D))
    nu_N, beta_N = C.c_double(), C.c_double()
    rc = lib().orc_hyper_params(D, C.c_int64(N), _dp(U), _dp(mu0), C.c_double(b0), _dp(Tinv), C.c_double(nu),
                                _dp(mu_N), _dp(T_N), C.byref(nu_N), C.byref(beta_N))
    if rc:
        raise np.linalg.LinAlgError("singular")
    return mu_N, beta_N.value, T_N.T.copy(), nu_N.value


# which map from the D mean normals to mu hyper_draw applies by default: "factor" -- through the factor Z of Lambda = Z Z' the Wishart
# draw holds (the library's default since round 5; the same law as the reference's, another function of z) -- or "reference" --
# chol(inv(Lambda) / beta_N)' z (normal_wishart.jl:40; the library with BDF_HYPER_MEAN=reference)
HYPER_MEAN_MAP = "factor"


def hyper_draw(mu_N, beta_N, T_N, nu_N, seed, sweep, entity_tag, mean_map=None):
    """rand(::NormalWishart) (normal_wishart.jl:38-42) -> mu (D), Lambda (D, D)"""
    mu_N = _f64(mu_N)
    D = len(mu_N)
    T = np.asfortranarray(_f64(T_N))
    mu = np.zeros(D)
    Lam = np.zeros((D, D))
    mm = {"reference": 0, "factor": 1}[mean_map or HYPER_MEAN_MAP]
    rc = lib().orc_hyper_draw2(D, _dp(mu_N), C.c_double(beta_N), T.ctypes.data_as(c_dp), C.c_double(nu_N),
                               C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(entity_tag), C.c_int(mm), _dp(mu), _dp(Lam))
    if rc:
        raise np.linalg.LinAlgError("not positive definite")
    return mu, Lam.T.copy()


def sample_lambda_beta(beta, Lambda, nu, mu, seed, sweep, entity_tag):
    beta = np.asfortranarray(_f64(beta))
    numF, D = beta.shape
    Lambda = _f64(Lambda)
    return lib().orc_sample_lambda_beta(D, C.c_int64(numF), beta.ctypes.data_as(c_dp), _dp(Lambda), C.c_double(nu),
                                        C.c_double(mu), C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(entity_tag))


def sample_alpha(alpha_lambda0, alpha_nu0, n, sumsq_err, seed, sweep, rel_tag):
    return lib().orc_sample_alpha(C.c_double(alpha_lambda0), C.c_double(alpha_nu0), C.c_int64(n),
                                  C.c_double(sumsq_err), C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(rel_tag))


# ---------------------------------------------------------------------------------------
class Feat:
    """Entity.F operator. kinds: 'dense' (N x numF array), 'csr' (scipy-like rows/cols/vals),
    'bincsr' (SparseBinMatrixCSR), 'bincoo' (SparseBinMatrix). 0-based indices."""

    def __init__(self, kind, m, n, dense=None, rowptr=None, colind=None, vals=None, rows=None, cols=None):
        self.kind, self.m, self.n = kind, int(m), int(n)
        self.dense = None if dense is None else np.asfortranarray(_f64(dense))
        self.rowptr = None if rowptr is None else np.ascontiguousarray(rowptr, dtype=np.int64)
        self.colind = None if colind is None else np.ascontiguousarray(colind, dtype=np.int32)
        self.vals = None if vals is None else _f64(vals)
        self.rows = None if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
        self.cols = None if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
        self.nnz = 0 if self.rows is None else len(self.rows)

    @staticmethod
    def from_dense(F):
        F = np.asarray(F, dtype=np.float64)
        return Feat("dense", F.shape[0], F.shape[1], dense=F)

    @staticmethod
    def _csr(rows, cols, m):
        rows = np.asarray(rows, dtype=np.int64)
        order = np.argsort(rows, kind="stable")          # sortperm(rows), sparsebin_csr.jl:23
        rp = np.zeros(m + 1, dtype=np.int64)
        np.add.at(rp, rows + 1, 1)
        return np.cumsum(rp), order

    @staticmethod
    def from_csr(rows, cols, vals, m, n):
        rp, order = Feat._csr(rows, cols, m)
        return Feat("csr", m, n, rowptr=rp, colind=np.asarray(cols)[order], vals=np.asarray(vals, dtype=np.float64)[order])

    @staticmethod
    def from_bincsr(rows, cols, m, n):
        rp, order = Feat._csr(rows, cols, m)
        return Feat("bincsr", m, n, rowptr=rp, colind=np.asarray(cols)[order])

    @staticmethod
    def from_bincoo(rows, cols, m, n):
        return Feat("bincoo", m, n, rows=rows, cols=cols)

    def struct(self):
        k = {"dense": 0, "csr": 1, "bincsr": 2, "bincoo": 3}[self.kind]
        g = lambda a, t: None if a is None else a.ctypes.data_as(t)
        return OrcFeat(k, self.m, self.n, self.nnz, g(self.dense, c_dp), g(self.rowptr, c_i64p), g(self.colind, c_i32p),
                       g(self.vals, c_dp), g(self.rows, c_i32p), g(self.cols, c_i32p))

    def mul(self, x):
        x = _f64(x)
        y = np.zeros(self.m)
        s = self.struct()
        lib().orc_feat_mul(C.byref(s), _dp(x), _dp(y))
        return y

    def tmul(self, x):
        x = _f64(x)
        y = np.zeros(self.n)
        s = self.struct()
        lib().orc_feat_tmul(C.byref(s), _dp(x), _dp(y))
        return y

    def AtA_mul_B(self, x, lam):
        x = _f64(x)
        y = np.zeros(self.n)
        tmp = np.zeros(self.m)
        s = self.struct()
        lib().orc_AtA_mul_B(C.byref(s), _dp(x), C.c_double(lam), _dp(y), _dp(tmp))
        return y

    def cg_AtA(self, b, lam, tol=None, maxiter=None):
        b = _f64(b)
        x = np.zeros(self.n)
        s = self.struct()
        tol = self.n * np.finfo(np.float64).eps if tol is None else tol
        maxiter = self.n if maxiter is None else maxiter
        it = lib().orc_cg_AtA(C.byref(s), _dp(b), C.c_double(lam), C.c_double(tol), int(maxiter), _dp(x))
        return x, it


def solve_full(FF, rhs, lam):
    FF = np.asfortranarray(_f64(FF))
    rhs2 = np.asfortranarray(_f64(rhs).reshape(FF.shape[0], -1))
    out = np.zeros_like(rhs2, order="F")
    rc = lib().orc_solve_full(C.c_int64(FF.shape[0]), FF.ctypes.data_as(c_dp), rhs2.ctypes.data_as(c_dp),
                              rhs2.shape[1], C.c_double(lam), out.ctypes.data_as(c_dp))
    if rc:
        raise np.linalg.LinAlgError("singular")
    return out.reshape(np.shape(rhs))


def sample_beta(feat, sample, mu, Lambda, lambda_beta, use_ff, tol, seed, sweep, entity_tag, maxiter=0):
    """sample_beta (sampling.jl:291-312). sample: (N, D). Returns beta (numF, D), rhs (numF, D), iters (D)"""
    sample, mu, Lambda = _f64(sample), _f64(mu), _f64(Lambda)
    N, D = sample.shape
    beta = np.zeros((feat.n, D), order="F")
    rhs = np.zeros((feat.n, D), order="F")
    iters = np.zeros(D, dtype=np.int32)
    s = feat.struct()
    rc = lib().orc_sample_beta(C.byref(s), D, _dp(sample), _dp(mu), _dp(Lambda), C.c_double(lambda_beta),
                               int(bool(use_ff)), C.c_double(np.nan if tol is None else tol), int(maxiter),
                               C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(entity_tag),
                               beta.ctypes.data_as(c_dp), rhs.ctypes.data_as(c_dp), iters.ctypes.data_as(C.POINTER(C.c_int)))
    if rc:
        raise np.linalg.LinAlgError("sample_beta failed")
    return beta, rhs, iters


def sample_beta_rel(feat, res, alpha, lambda_beta, seed, sweep, rel_tag):
    """sample_beta_rel (sampling.jl:322-337), FF path. res: values - udot - mean_value per observation. Returns beta, rhs"""
    res = _f64(res)
    beta, rhs = np.zeros(feat.n), np.zeros(feat.n)
    s = feat.struct()
    rc = lib().orc_sample_beta_rel(C.byref(s), _dp(res), C.c_double(alpha), C.c_double(lambda_beta), C.c_uint64(seed),
                                   C.c_uint32(sweep), C.c_uint32(rel_tag), beta.ctypes.data_as(c_dp), rhs.ctypes.data_as(c_dp))
    if rc:
        raise np.linalg.LinAlgError("sample_beta_rel failed")
    return beta, rhs


def noise_rows(D, n, Lambda, seed, sweep, purpose, entity_tag):
    Lambda = _f64(Lambda)
    E = np.zeros((n, D), order="F")
    rc = lib().orc_noise_rows(D, C.c_int64(n), _dp(Lambda), C.c_uint64(seed), C.c_uint32(sweep), C.c_uint32(purpose),
                              C.c_uint32(entity_tag), E.ctypes.data_as(c_dp))
    if rc:
        raise np.linalg.LinAlgError("Lambda not positive definite")
    return E


def predict(ids, factors, mean_value):
    ids = np.asfortranarray(np.asarray(ids, dtype=np.int64))
    n, n_modes = ids.shape
    fs = [_f64(f) for f in factors]
    D = fs[0].shape[1]
    fp = (c_dp * n_modes)(*[_dp(f) for f in fs])
    out = np.zeros(n)
    lib().orc_predict(D, n_modes, C.c_int64(n), ids.ctypes.data_as(c_i64p), fp, C.c_double(mean_value), _dp(out))
    return out


def num_threads():
    return lib().orc_num_threads()
