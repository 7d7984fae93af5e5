"""IndexedDF / FastIDF -- host mirror of src/IndexedDF.jl of the reference.

The table is held as an integer id matrix (nnz x n_modes, 1-based like the reference's DataFrame) and a value
vector; the per-mode adjacency index is built by the C library (`bdf_index_build`, host code of libbdf_hip.so)
in the reference's order (IndexedDF.jl:10-19: row numbers pushed in table order).  Accessors take 1-based mode
and entity numbers exactly like the Julia functions they mirror, so the reference's tests read the same here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError, c_i64p


def _split_table(df):
    """Accept a pandas DataFrame (N-1 id columns + 1 value column), a 2-tuple (ids, values) or a dict."""
    names = None
    if hasattr(df, "columns") and hasattr(df, "iloc"):          # pandas
        names = [str(c) for c in df.columns]
        ids = df.iloc[:, :-1].to_numpy()
        vals = df.iloc[:, -1].to_numpy()
    elif isinstance(df, dict):
        names = list(df.keys())
        cols = [np.asarray(df[k]) for k in names]
        ids = np.stack(cols[:-1], axis=1) if len(cols[0]) else np.zeros((0, len(cols) - 1), dtype=np.int64)
        vals = cols[-1]
    else:
        ids, vals = df
        ids = np.asarray(ids)
        vals = np.asarray(vals)
    if ids.ndim == 1:
        ids = ids.reshape(-1, 1)
    return ids, vals, names


class IndexedDF:
    """IndexedDF(df, dims) (IndexedDF.jl:6-24).

    df   : pandas DataFrame / dict of columns / (ids, values); ids are 1-based.
    dims : list or tuple of entity counts; default = column maxima (IndexedDF.jl:24).
    Integer ids may be int32 or int64 and values float32 or float64 (FastIDF{Ti,Tv}, test/basic.jl:32-33).
    """

    def __init__(self, df, dims=None, names=None):
        ids, vals, nm = _split_table(df)
        if not np.issubdtype(ids.dtype, np.integer):
            if ids.size and not np.all(ids == np.floor(ids)):
                raise ArgumentError("id columns must hold integers")
            ids = ids.astype(np.int64)
        if ids.dtype not in (np.int32, np.int64):
            ids = ids.astype(np.int64)
        if vals.dtype not in (np.float32, np.float64):
            vals = vals.astype(np.float64)
        self.ids = np.asfortranarray(ids)
        self.values = np.ascontiguousarray(vals)
        self.names = names or nm or [f"E{i + 1}" for i in range(ids.shape[1])] + ["value"]
        n_modes = self.ids.shape[1]
        if dims is None:
            dims = [int(self.ids[:, m].max()) if len(self.ids) else 0 for m in range(n_modes)]
        self.dims = [int(d) for d in dims]
        if len(self.dims) != n_modes:
            raise ArgumentError(f"dims has {len(self.dims)} entries but the table has {n_modes} id columns")
        self._index = None          # built on first use: a 100M-row relation that only goes to the device never needs it here
        if self.ids.shape[0] <= (1 << 22):
            self._build_index()     # (small tables: at once, so that an id outside 1..dims raises here as in the reference)
        else:
            for m, d in enumerate(self.dims):
                col = self.ids[:, m]
                if col.size and (int(col.min()) < 1 or int(col.max()) > d):
                    raise _lib.BoundsError(f"id of mode {m + 1} outside 1..{d} (BoundsError)")

    @property
    def _rowptr(self):
        if self._index is None:
            self._build_index()
        return self._index[0]

    @property
    def _rowids(self):
        if self._index is None:
            self._build_index()
        return self._index[1]

    def _build_index(self):
        nnz, n_modes = self.ids.shape
        dims = np.asarray(self.dims, dtype=np.int64)
        rowptr = [np.zeros(d + 1, dtype=np.int64) for d in self.dims]
        rowids = [np.zeros(max(nnz, 1), dtype=np.int64) for _ in self.dims]
        rp = (c_i64p * n_modes)(*[a.ctypes.data_as(c_i64p) for a in rowptr])
        ri = (c_i64p * n_modes)(*[a.ctypes.data_as(c_i64p) for a in rowids])
        _lib.check(_lib.lib().bdf_index_build(n_modes, dims.ctypes.data_as(c_i64p), nnz, self.ids.ctypes.data_as(C.c_void_p),
                                              self.ids.dtype.itemsize, rp, ri))
        self._index = (rowptr, [r[:nnz] for r in rowids])

    # ---- the reference's accessors (1-based mode / entity numbers) ------------------------------------------
    @property
    def index(self):
        """index[mode-1][j-1] -> 1-based row numbers (Vector{Vector{Vector{Int64}}}, IndexedDF.jl:8)"""
        return [[self._rowids[m][self._rowptr[m][j]:self._rowptr[m][j + 1]] for j in range(self.dims[m])]
                for m in range(len(self.dims))]

    def nnz(self):
        return self.ids.shape[0]

    def size(self, i=None):
        return tuple(self.dims) if i is None else self.dims[i - 1]

    def valueMean(self):
        return float(np.mean(self.values.astype(np.float64)))

    def getI(self, mode, i):
        m = mode - 1
        if not (1 <= i <= self.dims[m]):
            raise _lib.BoundsError(f"entity {i} outside 1..{self.dims[m]}")
        return self._rowids[m][self._rowptr[m][i - 1]:self._rowptr[m][i]]

    def getCount(self, mode, i):
        return int(len(self.getI(mode, i)))

    def getData(self, mode, i):
        """rows of the table whose id in `mode` is i -> (ids_sub, values_sub) (IndexedDF.jl:41, 67-70)"""
        rows = self.getI(mode, i) - 1
        return self.ids[rows, :], self.values[rows]

    def getValues(self):
        return self.values.astype(np.float64)

    def getMode(self, mode):
        return self.ids[:, mode - 1]

    def removeSamples(self, samples):
        """removeSamples(idf, samples) (IndexedDF.jl:34-37): samples are 1-based row numbers"""
        keep = np.ones(self.nnz(), dtype=bool)
        keep[np.asarray(samples, dtype=np.int64) - 1] = False
        return IndexedDF((self.ids[keep, :], self.values[keep]), self.dims, names=self.names)


class FastIDF:
    """FastIDF (IndexedDF.jl:46-70): the same index over plain id / value arrays."""

    def __init__(self, idf, dims=None):
        if not isinstance(idf, IndexedDF):
            idf = IndexedDF(idf, dims)
        self.ids, self.values, self._idf = idf.ids, idf.values, idf
        self.Ti, self.Tv = self.ids.dtype, self.values.dtype

    @property
    def index(self):
        return self._idf.index

    def getData(self, mode, i):
        return self._idf.getData(mode, i)

    def size(self, i=None):
        return self._idf.size(i)

    def nnz(self):
        return self._idf.nnz()


def nnz(x):
    return x.nnz()


def getData(x, mode, i):
    return x.getData(mode, i)


def getCount(x, mode, i):
    return x.getCount(mode, i)


def getI(x, mode, i):
    return x.getI(mode, i)


def getValues(x):
    return x.getValues()


def valueMean(x):
    return x.valueMean()


def removeSamples(x, samples):
    return x.removeSamples(samples)
