"""bayesiandatafusion.jl_amd -- MI355X-native Gibbs sweep behind the Entity / Relation / RelationData / macau(...)
surface of jaak-s/BayesianDataFusion.jl.

The compute lives in csrc/libbdf_hip.so (hand-written HIP for gfx950, C ABI in include/bdf.h).  This package is the
host-side mirror of the reference's data model and driver; it holds no numeric fallback.

The directory name contains a dot, so import it through the root-level shim:  `import bdf_amd`.
"""
from ._lib import (ArgumentError, BoundsError, DimensionMismatch, HipError, NoGpuError, NotPositiveDefinite,
                   declared_symbols, lib, LIB_PATH)
from .indexed_df import (IndexedDF, FastIDF, nnz, getData, getCount, getI, getValues, valueMean, removeSamples)
from .features import SparseBinMatrix, SparseBinMatrixCSR, SparseMatrixCSR, sparse_csr
from .relation_data import (Entity, EntityModel, Relation, RelationModel, RelationData, addRelation, assignToTest, setTest,
                            setPrecision, numData, numTest, hasFeatures, toStr, normalizeFeatures, normalizeRows)
from .data_reading import (read_ecfp, read_sparse, read_rowcol, read_binary_int32, filter_rare, write_binary_int32,
                           write_binary_matrix, read_binary_float32, read_sparse_float32, write_sparse_float32,
                           read_sparse_float64, write_sparse_float64, read_sparse_binary_matrix, write_sparse_binary_matrix,
                           read_matrix_market, write_matrix_market)


def rep_int(x, times):
    """rep_int (src/RelationData.jl:283-291)"""
    import numpy as np
    return np.repeat(np.asarray(x), np.asarray(times))


def __getattr__(name):
    # the engine and driver import torch; load them on first use so that the host-only data model (and the
    # CPU test suite) does not pay for it
    if name in ("macau", "pred", "pred_all", "AUC_ROC", "makeClamped"):
        import importlib
        return getattr(importlib.import_module(__name__ + ".driver"), name)
    if name in ("Block", "sample_users_blocked"):
        import importlib
        return getattr(importlib.import_module(__name__ + ".blocked"), name)
    if name in ("GibbsEngine", "Context", "DeviceRelation", "DevicePairs", "FeatOperator"):
        import importlib
        return getattr(importlib.import_module(__name__ + ".engine"), name)
    raise AttributeError(name)
