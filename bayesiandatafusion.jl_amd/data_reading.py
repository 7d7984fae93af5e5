"""On-disk formats of the reference (src/data_reading.jl): host-side readers and writers, byte layouts as Julia's
`write(f, ...)` produces them (little-endian, Int64 headers, column-major payloads).  Downstream tools and the feature
loaders of the reference's examples consume these files; nothing here touches the GPU."""
import numpy as np
import scipy.sparse as sp

__all__ = ["read_ecfp", "read_sparse", "read_rowcol", "read_binary_int32", "filter_rare", "write_binary_int32",
           "write_binary_matrix", "read_binary_float32", "read_sparse_float32", "write_sparse_float32", "read_sparse_float64",
           "write_sparse_float64", "read_sparse_binary_matrix", "write_sparse_binary_matrix", "read_matrix_market",
           "write_matrix_market"]


def _findnz(X):
    """findnz of a SparseMatrixCSC: 1-based (rows, cols, values) in column-major order"""
    X = sp.csc_matrix(X)
    X.sort_indices()
    cols = np.repeat(np.arange(X.shape[1]), np.diff(X.indptr))
    return X.indices.astype(np.int64) + 1, cols.astype(np.int64) + 1, X.data


def read_ecfp(filename):
    """data_reading.jl:10-38: CSV lines `id,fp,fp,...`; fingerprints renumbered 1.. in order of first appearance.
    Returns rows, cols (Int32, 1-based) and the raw -> id dictionary"""
    fp, rows, cols = {}, [], []
    i = 0
    with open(filename) as f:
        for line in f:
            i += 1
            a = line.rstrip("\n").split(",")
            for tok in a[1:]:
                raw = int(tok)
                if raw not in fp:
                    fp[raw] = len(fp) + 1
                rows.append(i)
                cols.append(fp[raw])
    return np.asarray(rows, dtype=np.int32), np.asarray(cols, dtype=np.int32), fp


def read_rowcol(filename):
    """data_reading.jl:40-51"""
    a = np.loadtxt(filename, delimiter=",", dtype=np.int64, usecols=(0, 1), ndmin=2)
    return a[:, 0].astype(np.int32), a[:, 1].astype(np.int32)


def _read_dense(filename, dtype):
    with open(filename, "rb") as f:
        nrows, ncols = np.fromfile(f, dtype="<i8", count=2)
        return np.fromfile(f, dtype=dtype, count=int(nrows * ncols)).reshape((int(ncols), int(nrows))).T.copy()


def read_binary_int32(filename):
    """data_reading.jl:53-59: Int64 nrows, Int64 ncols, Int32 column-major"""
    return _read_dense(filename, "<i4")


def read_binary_float32(filename):
    """data_reading.jl:61-67: Int64 nrows, Int64 ncols, Float32 column-major"""
    return _read_dense(filename, "<f4")


def write_binary_matrix(filename, X):
    """data_reading.jl:93-99: Int64 nrows, Int64 ncols, then X column-major in its own element type"""
    X = np.asarray(X)
    with open(filename, "wb") as f:
        np.asarray(X.shape[:2] if X.ndim == 2 else (X.shape[0], 1), dtype="<i8").tofile(f)
        np.asfortranarray(X).T.tofile(f)


def write_binary_int32(filename, X):
    """data_reading.jl:89-91"""
    write_binary_matrix(filename, np.asarray(X, dtype=np.int32))


def read_sparse_float32(filename):
    """data_reading.jl:69-77: Int64 nnz, Int32 rows, Int32 cols, Float32 values -> (rows, cols, vals)"""
    with open(filename, "rb") as f:
        nnz = int(np.fromfile(f, dtype="<i8", count=1)[0])
        return np.fromfile(f, "<i4", nnz), np.fromfile(f, "<i4", nnz), np.fromfile(f, "<f4", nnz)


def write_sparse_float32(filename, X_or_rows, cols=None, values=None):
    """data_reading.jl:101-120: a sparse matrix (findnz order) or explicit (rows, cols, values)"""
    if cols is None:
        rows, cols, values = _findnz(X_or_rows)
    else:
        rows = X_or_rows
    with open(filename, "wb") as f:
        np.asarray([len(rows)], dtype="<i8").tofile(f)
        np.asarray(rows, dtype="<i4").tofile(f)
        np.asarray(cols, dtype="<i4").tofile(f)
        np.asarray(values, dtype="<f4").tofile(f)


def read_sparse(filename):
    """data_reading.jl:79-82: rows,cols CSV -> sparse matrix of ones (duplicates add up, as sparse() does)"""
    rows, cols = read_rowcol(filename)
    return sp.csc_matrix((np.ones(len(rows), dtype=np.float32), (rows - 1, cols - 1)))


def filter_rare(X, nmin):
    """data_reading.jl:84-87: keep the columns whose sum is at least nmin"""
    X = sp.csc_matrix(X)
    featn = np.asarray(X.sum(axis=0)).ravel()
    return X[:, featn >= nmin]


def write_sparse_binary_matrix(filename, X):
    """data_reading.jl:122-132: Int64 nrows, ncols, nnz, then Int32 rows, Int32 cols of the non-zeros"""
    rows, cols, _ = _findnz(X)
    with open(filename, "wb") as f:
        np.asarray([X.shape[0], X.shape[1], len(rows)], dtype="<i8").tofile(f)
        rows.astype("<i4").tofile(f)
        cols.astype("<i4").tofile(f)


def read_sparse_binary_matrix(filename):
    """data_reading.jl:134-143"""
    with open(filename, "rb") as f:
        nrows, ncols, nnz = (int(x) for x in np.fromfile(f, dtype="<i8", count=3))
        rows, cols = np.fromfile(f, "<i4", nnz), np.fromfile(f, "<i4", nnz)
    return sp.csc_matrix((np.ones(nnz, dtype=np.int64), (rows - 1, cols - 1)), shape=(nrows, ncols))


def write_sparse_float64(filename, X):
    """data_reading.jl:195-206: Int64 nrow, ncol, nnz, Int32 rows, Int32 cols, Float64 values"""
    rows, cols, vals = _findnz(X)
    with open(filename, "wb") as f:
        np.asarray([X.shape[0], X.shape[1], len(rows)], dtype="<i8").tofile(f)
        rows.astype("<i4").tofile(f)
        cols.astype("<i4").tofile(f)
        np.asarray(vals, dtype="<f8").tofile(f)


def read_sparse_float64(filename):
    """data_reading.jl:208-218"""
    with open(filename, "rb") as f:
        nrow, ncol, nnz = (int(x) for x in np.fromfile(f, dtype="<i8", count=3))
        rows, cols, vals = np.fromfile(f, "<i4", nnz), np.fromfile(f, "<i4", nnz), np.fromfile(f, "<f8", nnz)
    return sp.csc_matrix((vals, (rows - 1, cols - 1)), shape=(nrow, ncol))


def read_matrix_market(filename):
    """data_reading.jl:145-180: coordinate format, '%' comment lines, header `nrows ncols nnz`"""
    rows, cols, vals = [], [], []
    header = None
    with open(filename) as f:
        for ln in f:
            if not ln.strip() or ln[0] == "%":
                continue
            arr = ln.split()
            if header is None:
                header = (int(arr[0]), int(arr[1]), int(arr[2]))
                continue
            rows.append(int(arr[0])); cols.append(int(arr[1])); vals.append(float(arr[2]))
    nrows, ncols, _ = header
    return sp.csc_matrix((np.asarray(vals), (np.asarray(rows) - 1, np.asarray(cols) - 1)), shape=(nrows, ncols))


def write_matrix_market(filename, X):
    """data_reading.jl:182-193: X is a table whose first three columns are row, column, value"""
    A = np.asarray(X)[:, :3] if not hasattr(X, "iloc") else X.iloc[:, :3].to_numpy()
    with open(filename, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("%d\t%d\t%d\n" % (int(A[:, 0].max()), int(A[:, 1].max()), A.shape[0]))
        for r, c, v in A:
            f.write("%d\t%d\t%s\n" % (int(r), int(c), repr(float(v))))
