"""Host-side holders of the side-information matrix types the reference accepts as Entity.F.

  dense numpy array                      Matrix{Float64}
  scipy.sparse matrix / SparseMatrixCSR  SparseMatrixCSC / SparseMatrixCSR  (src/parallel_csr.jl:36-54)
  SparseBinMatrix                        binary COO, Int32                  (src/parallel_matrix.jl:9-46, 225-227)
  SparseBinMatrixCSR                     binary CSR, Int32                  (src/sparsebin_csr.jl:6-37)

These classes only hold and index the data (1-based rows/cols like the reference); products are computed by the
device operator `bdf_feat` (engine.FeatOperator).  The reference's ParallelSBM / ParallelBinCSR / psparse wrappers
distribute the same matrices over worker processes; here one device operator serves all of them.
"""
import numpy as np

from ._lib import ArgumentError, DimensionMismatch


class SparseBinMatrix:
    """SparseBinMatrix(rows, cols) / SparseBinMatrix(m, n, rows, cols) (parallel_matrix.jl:9-24)"""

    def __init__(self, *args):
        if len(args) == 2:
            rows, cols = args
            m = n = None
        elif len(args) == 4:
            m, n, rows, cols = args
        else:
            raise ArgumentError("SparseBinMatrix(rows, cols) or SparseBinMatrix(m, n, rows, cols)")
        rows = np.asarray(rows).astype(np.int32)
        cols = np.asarray(cols).astype(np.int32)
        if len(rows) != len(cols):
            raise DimensionMismatch("length(rows) must equal length(cols)")
        self.rows, self.cols = rows, cols
        self.m = int(rows.max()) if m is None else int(m)
        self.n = int(cols.max()) if n is None else int(n)

    @property
    def shape(self):
        return (self.m, self.n)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d - 1]

    def isempty(self):
        return self.m == 0 or self.n == 0

    def __getitem__(self, key):
        """boolean row subsetting, sbm[rows, :] (parallel_matrix.jl:27-45)"""
        rows = key[0] if isinstance(key, tuple) else key
        rows = np.asarray(rows, dtype=bool)
        if len(rows) != self.m:
            raise DimensionMismatch("length(rows) must equal size(sbm,1)")
        idx = rows[self.rows - 1]
        rsum = np.cumsum(rows)
        return SparseBinMatrix(int(rows.sum()), self.n, rsum[self.rows[idx] - 1].astype(np.int32), self.cols[idx])

    def toarray(self):
        A = np.zeros(self.shape)
        np.add.at(A, (self.rows - 1, self.cols - 1), 1.0)
        return A


class SparseBinMatrixCSR(SparseBinMatrix):
    """SparseBinMatrixCSR(rows, cols) (sparsebin_csr.jl:22-37): same content, CSR layout on the device"""

    def __init__(self, rows, cols):
        super().__init__(rows, cols)
        order = np.argsort(self.rows, kind="stable")             # sortperm(rows)
        self.col_ind = self.cols[order]
        rp = np.zeros(self.m + 1, dtype=np.int64)
        np.add.at(rp, self.rows, 1)
        self.row_ptr = (np.cumsum(rp) + 1).astype(np.int32)      # 1-based like the reference


class SparseMatrixCSR:
    """sparse_csr(rows, cols, vals) / sparse_csr(csc) (parallel_csr.jl:36-41)"""

    def __init__(self, rows, cols, vals, m=None, n=None):
        self.rows = np.asarray(rows).astype(np.int32)
        self.cols = np.asarray(cols).astype(np.int32)
        self.vals = np.asarray(vals, dtype=np.float64)
        self.m = int(self.rows.max()) if m is None else int(m)
        self.n = int(self.cols.max()) if n is None else int(n)

    @property
    def shape(self):
        return (self.m, self.n)

    def size(self, d=None):
        return self.shape if d is None else (1 if d > 2 else self.shape[d - 1])

    def isempty(self):
        return self.m == 0 or self.n == 0

    def toarray(self):
        A = np.zeros(self.shape)
        np.add.at(A, (self.rows - 1, self.cols - 1), self.vals)
        return A


def sparse_csr(*args):
    if len(args) == 1:
        coo = args[0].tocoo()
        return SparseMatrixCSR(coo.row + 1, coo.col + 1, coo.data, coo.shape[0], coo.shape[1])
    return SparseMatrixCSR(*args)


def feature_shape(F):
    if F is None:
        return (0, 0)
    if hasattr(F, "shape"):
        return tuple(int(x) for x in F.shape)
    raise ArgumentError(f"unsupported feature matrix type {type(F)}")


def isempty(F):
    m, n = feature_shape(F)
    return m == 0 or n == 0


def subset_rows(F, keep):
    """F[keep, :] for any supported type (assignToTest!, RelationData.jl:204-209)"""
    keep = np.asarray(keep, dtype=bool)
    if isinstance(F, SparseBinMatrix):
        return F[keep, :]
    if isinstance(F, SparseMatrixCSR):
        idx = keep[F.rows - 1]
        rsum = np.cumsum(keep)
        return SparseMatrixCSR(rsum[F.rows[idx] - 1], F.cols[idx], F.vals[idx], int(keep.sum()), F.n)
    if hasattr(F, "tocsr"):
        return F.tocsr()[np.nonzero(keep)[0], :]
    return np.asarray(F)[keep, :]


def take_rows(F, rows0):
    """F[rows, :] with 0-based integer rows, order kept (r.test_F = r.F[test_id,:])"""
    if hasattr(F, "tocsr"):
        return F.tocsr()[rows0, :]
    if isinstance(F, (SparseBinMatrix, SparseMatrixCSR)):
        return np.asarray(F.toarray())[rows0, :]
    return np.asarray(F)[rows0, :]
