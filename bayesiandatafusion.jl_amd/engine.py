"""GibbsEngine -- host-side orchestration of one macau() run on one MI355X (one process per GPU).

Everything numeric is a call into libbdf_hip.so (include/bdf.h).  torch is used for what the task calls plumbing: device
allocations (tensors) and, with several ranks, the channel over which rank 0 hands the RCCL unique id to the others.

* Without side information the whole iteration (rows of every entity, exchange between the GPUs, hyperpriors, test
  prediction update) is ONE native call, bdf_gibbs_sweep (csrc/bdf_gibbs.hip).  With side information the same steps plus
  uhat / beta are enqueued from here through the entry points of bdf.h, on the same three-stream schedule.
* Several ranks (shard=(rank, world)): every entity's rows are shared out by bdf_layout_build -- dealt over the ranks in
  falling order of degree as the reference deals rows i:P:N to its workers (src/sampling.jl:154) -- and stored at INTERNAL
  positions, so that a rank's rows of a chunk are one contiguous block and the exchange is an in-place all-gather
  (bdf_allgather_rows: RCCL over xGMI).  A rank holds the observations of its own rows only
  (bdf_relation_create_sharded) and a replica of every factor matrix.  Random streams are keyed by the original row id.

Layout: an entity's sample is the reference's D x N column-major matrix == a contiguous torch tensor of shape (N, D)
(N = chunks * world * cmax rows with a layout; EntityState.host() returns the reference's orientation and order).
"""
import ctypes as C
import os
import sys
import weakref

import numpy as np
import torch

from . import _lib
from . import features as feat
from ._lib import ArgumentError, GibbsEntity, Term, check, lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Layout:
    """internal row positions of an entity (bdf_layout_build); world == 1: the identity"""

    def __init__(self, N, degree=None, world=1, chunks=1):
        self.N, self.world, self.chunks = int(N), int(world), int(chunks)
        if world == 1 and chunks == 1:
            self.pos, self.cmax, self.nint = None, self.N, self.N
            return
        degree = np.ascontiguousarray(degree, dtype=np.int64)
        self.pos = np.zeros(self.N, dtype=np.int32)
        cmax = C.c_int64(0)
        check(lib().bdf_layout_build(self.N, degree.ctypes.data_as(_lib.c_i64p), world, chunks,
                                     self.pos.ctypes.data_as(_lib.c_i32p), C.byref(cmax)))
        self.cmax = cmax.value
        self.nint = self.chunks * self.world * self.cmax

    def to_internal(self, ids1):
        """1-based original ids -> 1-based internal positions"""
        ids1 = np.asarray(ids1)
        return ids1 if self.pos is None else self.pos[ids1 - 1].astype(np.int64) + 1

    def block(self, rank, chunk):
        """[begin, end) of the rows of (rank, chunk) in the factor matrix"""
        b = (chunk * self.world + rank) * self.cmax
        return b, b + self.cmax


class KernelTimer:
    """a pair of HIP events that bdf_ctx_time_next_rows attaches to the next row-kernel dispatch (the kernel's own begin and
    end on its stream; events recorded around the launch would add their marker packets to the interval)"""

    def __init__(self):
        self.start, self.stop = C.c_void_p(), C.c_void_p()
        check(lib().bdf_event_create(C.byref(self.start)))
        check(lib().bdf_event_create(C.byref(self.stop)))

    def elapsed_us(self):
        us = C.c_double(0.0)
        check(lib().bdf_event_elapsed_us(self.start, self.stop, C.byref(us)))
        return us.value

    def __del__(self):
        try:
            lib().bdf_event_destroy(self.start)
            lib().bdf_event_destroy(self.stop)
        except Exception:
            pass


class Context:
    """bdf_ctx bound to a torch device and torch's current stream."""

    def __init__(self, device=None, seed=0, stream=None):
        if not torch.cuda.is_available():
            raise _lib.NoGpuError("no GPU visible: bayesiandatafusion.jl_amd has no CPU path (the reference is the CPU path)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        torch.cuda.set_device(self.device)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.handle = C.c_void_p()
        check(lib().bdf_ctx_create(self.device.index, C.c_void_p(self.stream.cuda_stream), C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                   C.byref(self.handle)))
        self.seed = int(seed)
        # device objects created on this context; they are destroyed before it whatever order the garbage collector
        # finalises things in (a relation / feature handle points back into the context)
        self._children = weakref.WeakSet()

    @classmethod
    def wrap(cls, handle, device, seed, owned=False):
        """a context created by the library (bdf_ctx_create_side, bdf_gibbs_contexts): the stream is the library's"""
        self = cls.__new__(cls)
        self.device = device
        self.handle = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        st = C.c_void_p()
        check(lib().bdf_ctx_stream(self.handle, C.byref(st)))
        self.stream = torch.cuda.ExternalStream(st.value or 0, device=device)
        self.seed = int(seed)
        self._children = weakref.WeakSet()
        self._owned = owned
        return self

    @classmethod
    def rows(cls, device, seed, reserve_cus=0):
        """a row context on a library-owned stream that keeps `reserve_cus` CUs free for a side context (bdf_ctx_create_rows)"""
        if not torch.cuda.is_available():
            raise _lib.NoGpuError("no GPU visible: bayesiandatafusion.jl_amd has no CPU path (the reference is the CPU path)")
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        torch.cuda.set_device(dev)
        h = C.c_void_p()
        check(lib().bdf_ctx_create_rows(dev.index, C.c_uint64(int(seed) & (2 ** 64 - 1)), int(reserve_cus), C.byref(h)))
        return cls.wrap(h, dev, seed, owned=True)

    @classmethod
    def side(cls, main, apart=(), reserved=False):
        """a context on another stream of main's device, chosen by the library so that kernels on it really run beside those
        of `main` and of the contexts in `apart` (HIP multiplexes streams onto a few hardware queues; two streams on one
        queue serialise each other: 171 instead of 125 us per sweep); reserved: on the CUs main's stream leaves free"""
        h = C.c_void_p()
        arr = (C.c_void_p * max(len(apart), 1))(*[a.handle for a in apart])
        check(lib().bdf_ctx_create_side(main.handle, arr, len(apart), int(reserved), C.byref(h)))
        return cls.wrap(h, main.device, main.seed, owned=True)

    def adopt(self, child):
        self._children.add(child)

    def set_item_size(self, observations):
        check(lib().bdf_ctx_set_item_size(self.handle, int(observations)))

    def rows_unfinished(self):
        """parity hook: split rows left unfinished by the row-kernel launches so far (must be 0)"""
        n = C.c_int64(0)
        check(lib().bdf_rows_unfinished(self.handle, C.byref(n)))
        return n.value

    def set_small_rows(self, max_observations, min_rows):
        """D <= 16: rows of at most max_observations observations four to a wave when the entity has min_rows rows or more
        (bdf_ctx_set_small_rows; 0 observations: off)"""
        check(lib().bdf_ctx_set_small_rows(self.handle, int(max_observations), int(min_rows)))

    def set_lowrank(self, max_observations=-1, min_rows=8192):
        """D > 16: rows of at most max_observations observations (-1: D / 2 up to 16, up to 32 at D > 32; 0: off) by the low-rank sampler when a
        launch has min_rows such rows or more (bdf_ctx_set_lowrank) -- the same conditional distribution as the reference's
        map, other sampled values"""
        check(lib().bdf_ctx_set_lowrank(self.handle, int(max_observations), int(min_rows)))

    def set_col_rows(self, max_piece=-1):
        """16 < D <= 32, one two-mode relation: the rows four to a wave in the column layout, cut into pieces of at most max_piece
        observations (bdf_ctx_set_col_rows; 0: off -- the wave-per-row kernel; -1: the default, 128 unless the caller chose an item size)"""
        check(lib().bdf_ctx_set_col_rows(self.handle, int(max_piece)))

    def rows_dispatch(self, entity_tag):
        """how the latest row launch under entity_tag was dispatched on this context (bdf_ctx_rows_dispatch): a dict of rows by
        the low-rank sampler, k_rows_small, K1c and k_rows, k_rows' work items and K1c's waves; None before the first launch"""
        out = (C.c_int64 * 6)()
        if lib().bdf_ctx_rows_dispatch(self.handle, int(entity_tag), out) != 0:
            return None
        return dict(zip(("lowrank", "small", "col", "k1", "k1_items", "col_waves"), (int(x) for x in out)))

    def set_piece_size(self, observations):
        check(lib().bdf_ctx_set_piece_size(self.handle, int(observations)))

    def set_gather(self, mode):
        """parity hook: 0 auto, 1 general gather path, 2 lean path with 64-bit row offsets (num_latent > 32)"""
        check(lib().bdf_ctx_set_gather(self.handle, int(mode)))

    def set_sweep(self, i):
        check(lib().bdf_ctx_set_sweep(self.handle, C.c_uint32(int(i))))

    def advance_sweep(self):
        check(lib().bdf_ctx_advance_sweep(self.handle))

    def sync(self):
        check(lib().bdf_ctx_sync(self.handle))
        bits = C.c_uint32(0)
        check(lib().bdf_ctx_warnings(self.handle, C.byref(bits)))
        if bits.value & 64:          # BDF_WARN_CG_MAXITER: the reference returns such a column silently (parallel_cg.jl:73-93)
            import warnings
            warnings.warn("beta update: a conjugate-gradient column was still above its tolerance after maxiter iterations",
                          RuntimeWarning, stacklevel=2)

    def zeros(self, *shape, dtype=torch.float64):
        with torch.cuda.stream(self.stream):          # filled on the stream that will use it
            return torch.zeros(*shape, dtype=dtype, device=self.device)

    def tensor(self, a, dtype=torch.float64):
        with torch.cuda.stream(self.stream):
            return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def close(self):
        if self.handle:
            for ch in list(self._children):
                ch.close()
            if getattr(self, "_owned", True):      # (contexts of a bdf_gibbs object are destroyed with it)
                lib().bdf_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceRelation:
    """bdf_rel: Relation.data (IndexedDF) as per-mode CSR in HBM."""

    def __init__(self, ctx, idf, layouts=None, rank=0):
        """layouts: one Layout per mode (several ranks): the device then holds this rank's rows only, at internal positions"""
        self.ctx = ctx
        self.handle = C.c_void_p()
        dims = np.asarray(idf.dims, dtype=np.int64)
        vals = np.ascontiguousarray(idf.values, dtype=np.float64)
        ids = idf.ids
        if layouts is None or all(l.pos is None for l in layouts):
            check(lib().bdf_relation_create(ctx.handle, len(idf.dims), dims.ctypes.data_as(_lib.c_i64p), idf.nnz(),
                                            ids.ctypes.data_as(C.c_void_p), ids.dtype.itemsize, vals.ctypes.data_as(_lib.c_dp),
                                            C.byref(self.handle)))
        else:
            pos = (_lib.c_i32p * len(layouts))(*[l.pos.ctypes.data_as(_lib.c_i32p) for l in layouts])
            cmax = np.asarray([l.cmax for l in layouts], dtype=np.int64)
            check(lib().bdf_relation_create_sharded(ctx.handle, len(idf.dims), dims.ctypes.data_as(_lib.c_i64p), idf.nnz(),
                                                    ids.ctypes.data_as(C.c_void_p), ids.dtype.itemsize, vals.ctypes.data_as(_lib.c_dp),
                                                    pos, cmax.ctypes.data_as(_lib.c_i64p), rank, layouts[0].world,
                                                    layouts[0].chunks, C.byref(self.handle)))
        self.dims = list(idf.dims)
        self.nnz = idf.nnz()
        ctx.adopt(self)

    def index(self, mode0):
        rp, ri = _lib.c_i64p(), _lib.c_i64p()
        check(lib().bdf_relation_index(self.handle, mode0, C.byref(rp), C.byref(ri)))
        return (np.ctypeslib.as_array(rp, shape=(self.dims[mode0] + 1,)).copy(),
                np.ctypeslib.as_array(ri, shape=(max(self.nnz, 1),))[:self.nnz].copy())

    def value_mean(self):
        m = C.c_double()
        check(lib().bdf_relation_value_mean(self.handle, C.byref(m)))
        return m.value

    def order(self, mode0):
        out = np.zeros(self.dims[mode0], dtype=np.int32)
        check(lib().bdf_relation_order(self.handle, mode0, out.ctypes.data_as(_lib.c_i32p)))
        return out

    def close(self):
        if self.handle:
            if self.ctx.handle:                 # a closed context has already destroyed its objects
                lib().bdf_relation_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevicePairs:
    """bdf_pairs: test_vec (or the training table) with its running prediction state."""

    def __init__(self, ctx, ids, values):
        self.ctx = ctx
        self.handle = C.c_void_p()
        ids = np.asfortranarray(np.asarray(ids, dtype=np.int64))
        values = np.ascontiguousarray(values, dtype=np.float64)
        self.n, self.n_modes = ids.shape
        check(lib().bdf_pairs_create(ctx.handle, self.n_modes, self.n, ids.ctypes.data_as(C.c_void_p), 8,
                                     values.ctypes.data_as(_lib.c_dp), C.byref(self.handle)))
        self.stats = ctx.zeros(4)
        self._order = None
        ctx.adopt(self)

    def sort(self, mode0=0):
        """store the pairs sorted by their id in mode `mode0` (neighbouring pairs share that factor row); results keep the
        caller's order"""
        check(lib().bdf_pairs_sort(self.handle, int(mode0)))
        self._order = np.zeros(self.n, dtype=np.int64)
        check(lib().bdf_pairs_order(self.handle, self._order.ctypes.data_as(_lib.c_i64p)))
        return self

    def _facs(self, factors):
        return (C.c_void_p * len(factors))(*[f.data_ptr() for f in factors])

    def _on_own_stream(self):
        """the pairs' stream ordered after the caller's current stream (where the factors were produced), as a context
        manager under which torch allocates and fills on the pairs' stream"""
        cur = torch.cuda.current_stream(self.ctx.device)
        if cur != self.ctx.stream:
            self.ctx.stream.wait_stream(cur)
        return cur, torch.cuda.stream(self.ctx.stream)

    def _hand_back(self, cur, *tensors):
        """results written on the pairs' stream become visible to the caller's stream (a later .cpu() / kernel there)"""
        if cur != self.ctx.stream:
            cur.wait_stream(self.ctx.stream)
            for t in tensors:
                t.record_stream(cur)

    def predict(self, D, factors, mean_value):
        # the pairs of an engine live on its prediction stream (engine.test_pairs): `out` is allocated, zero-filled and
        # written there, and the caller's stream waits for it -- a plain ctx.zeros() here raced the side-stream kernel
        # against a null-stream fill and read-back (round-1 smoke)
        cur, own = self._on_own_stream()
        with own:
            out = torch.zeros(self.n, dtype=torch.float64, device=self.ctx.device)
            check(lib().bdf_predict(self.ctx.handle, self.handle, D, self._facs(factors), mean_value, _ptr(out)))
        self._hand_back(cur, out)
        return out

    def sse(self, D, factors, mean_value, linear=None):
        """device scalar (stats[1]): sum of (value - pred)^2, pred = udot + (linear | mean_value)"""
        check(lib().bdf_predict_sse(self.ctx.handle, self.handle, D, self._facs(factors), mean_value,
                                    _ptr(linear) if linear is not None else None, _ptr(self.stats)))
        return self.stats

    def update(self, D, factors, mean_value, phase, clamp, class_cut):
        lo, hi = (clamp[0], clamp[1]) if len(clamp) else (1.0, -1.0)
        check(lib().bdf_predict_update(self.ctx.handle, self.handle, D, self._facs(factors), mean_value, phase, lo, hi,
                                       class_cut, _ptr(self.stats)))
        return self.stats

    def state(self):
        """(avg, sq) as host arrays"""
        a, s, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(lib().bdf_pairs_state(self.handle, C.byref(a), C.byref(s), C.byref(n)))
        avg, sq = np.zeros(n.value), np.zeros(n.value)
        if n.value:
            check(lib().bdf_d2h(self.ctx.handle, avg.ctypes.data_as(C.c_void_p), a, n.value * 8))
            check(lib().bdf_d2h(self.ctx.handle, sq.ctypes.data_as(C.c_void_p), s, n.value * 8))
        if self._order is not None:             # storage order -> the caller's order
            a2, s2 = np.empty_like(avg), np.empty_like(sq)
            a2[self._order], s2[self._order] = avg, sq
            avg, sq = a2, s2
        return avg, sq

    def close(self):
        if self.handle:
            if self.ctx.handle:                 # a closed context has already destroyed its objects
                lib().bdf_pairs_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FeatOperator:
    """bdf_feat: the Entity.F operator on the device (dense / CSR / binary)."""

    def __init__(self, ctx, F, layout=None):
        """layout (several ranks): the rows of F move to the entity's internal positions (rows nobody owns are zero), and the
        original ids key the rows' noise streams (bdf_feat_set_row_ids)"""
        self.ctx = ctx
        self.handle = C.c_void_p()
        L = lib()
        self.m_orig = feat.feature_shape(F)[0]
        if layout is not None and layout.pos is not None:
            F = _rows_to_internal(F, layout)
        if isinstance(F, feat.SparseMatrixCSR):
            self.kind = "csr"
            check(L.bdf_feat_create_csr(ctx.handle, F.m, F.n, len(F.rows), F.rows.ctypes.data_as(_lib.c_i32p),
                                        F.cols.ctypes.data_as(_lib.c_i32p), F.vals.ctypes.data_as(_lib.c_dp), C.byref(self.handle)))
            self.m, self.n = F.m, F.n
        elif isinstance(F, feat.SparseBinMatrix):
            self.kind = "bin"
            check(L.bdf_feat_create_bin(ctx.handle, F.m, F.n, len(F.rows), F.rows.ctypes.data_as(_lib.c_i32p),
                                        F.cols.ctypes.data_as(_lib.c_i32p), C.byref(self.handle)))
            self.m, self.n = F.m, F.n
        elif hasattr(F, "tocoo"):
            coo = F.tocoo()
            rows = np.ascontiguousarray(coo.row + 1, dtype=np.int32)
            cols = np.ascontiguousarray(coo.col + 1, dtype=np.int32)
            vals = np.ascontiguousarray(coo.data, dtype=np.float64)
            self.kind = "csr"
            check(L.bdf_feat_create_csr(ctx.handle, coo.shape[0], coo.shape[1], len(rows), rows.ctypes.data_as(_lib.c_i32p),
                                        cols.ctypes.data_as(_lib.c_i32p), vals.ctypes.data_as(_lib.c_dp), C.byref(self.handle)))
            self.m, self.n = int(coo.shape[0]), int(coo.shape[1])
        else:
            A = np.asfortranarray(np.asarray(F, dtype=np.float64))
            if A.ndim != 2:
                raise ArgumentError("feature matrix must be two-dimensional")
            self.kind = "dense"
            check(L.bdf_feat_create_dense(ctx.handle, A.shape[0], A.shape[1], A.ctypes.data_as(_lib.c_dp), C.byref(self.handle)))
            self.m, self.n = int(A.shape[0]), int(A.shape[1])
        if layout is not None and layout.pos is not None:
            ids = np.full(layout.nint, -1, dtype=np.int32)
            ids[layout.pos] = np.arange(layout.N, dtype=np.int32)
            check(L.bdf_feat_set_row_ids(self.handle, ids.ctypes.data_as(_lib.c_i32p)))
        ctx.adopt(self)

    # B, out: torch tensors holding column-major matrices, i.e. shape (ncol, rows)
    def mul(self, B, transpose=False):
        ncol = B.shape[0]
        out = self.ctx.zeros(ncol, self.n if transpose else self.m)
        check(lib().bdf_feat_mul(self.ctx.handle, self.handle, _ptr(B), ncol, _ptr(out), int(transpose)))
        return out

    def AtA_mul(self, X, lam):
        out = self.ctx.zeros(X.shape[0], self.n)
        check(lib().bdf_feat_AtA_mul(self.ctx.handle, self.handle, _ptr(X), X.shape[0], float(lam), _ptr(out)))
        return out

    def close(self):
        if self.handle:
            if self.ctx.handle:                 # a closed context has already destroyed its objects
                lib().bdf_feat_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _rows_to_internal(F, layout):
    """F with row i moved to row layout.pos[i] of a layout.nint-row matrix of the same kind"""
    pos = layout.pos.astype(np.int64)
    if isinstance(F, feat.SparseMatrixCSR):
        return feat.SparseMatrixCSR(pos[F.rows - 1] + 1, F.cols, F.vals, layout.nint, F.n)
    if isinstance(F, feat.SparseBinMatrix):
        return feat.SparseBinMatrix(layout.nint, F.n, pos[F.rows - 1] + 1, F.cols)
    if hasattr(F, "tocoo"):
        import scipy.sparse as sp
        coo = F.tocoo()
        return sp.csr_matrix((coo.data, (pos[coo.row], coo.col)), shape=(layout.nint, coo.shape[1]))
    A = np.asarray(F, dtype=np.float64)
    out = np.zeros((layout.nint, A.shape[1]))
    out[pos] = A
    return out


def _take_rows(F, rows0):
    """the rows rows0 (0-based, in that order) of F as a matrix of the same kind"""
    rows0 = np.asarray(rows0, dtype=np.int64)
    m = feat.feature_shape(F)[0]
    if isinstance(F, (feat.SparseMatrixCSR, feat.SparseBinMatrix)):
        new = np.full(m, -1, dtype=np.int64)
        new[rows0] = np.arange(len(rows0))                    # (rows0 holds no row twice: blocks and test subsets)
        to = new[np.asarray(F.rows, dtype=np.int64) - 1]
        keep = to >= 0
        rr = (to[keep] + 1).astype(np.int32)
        if isinstance(F, feat.SparseMatrixCSR):
            return feat.SparseMatrixCSR(rr, F.cols[keep], F.vals[keep], len(rows0), F.n)
        return feat.SparseBinMatrix(len(rows0), F.n, rr, F.cols[keep])
    if hasattr(F, "tocsr"):
        return F.tocsr()[rows0]
    return np.asarray(F, dtype=np.float64)[rows0]


class EntityState:
    """Device-resident EntityModel (RelationData.jl:14-40, initModel! :66-90)."""

    def __init__(self, ctx, en, D, tag, layout):
        self.ctx, self.D, self.tag, self.layout = ctx, D, tag, layout
        self.n_real = en.count
        self.N = layout.nint                     # rows of the factor matrix (rows nobody owns stay zero)
        # the rows of the next sweep are written to another buffer, then the three rotate: a launch overwrites the rows of
        # three sweeps ago, so readers of the previous two sweeps' rows on other streams are never overwritten under their feet
        self.bufs = [ctx.zeros(self.N, D) for _ in range(3)]
        self.cur = 0
        self.mu = ctx.zeros(D)
        self.Lambda = ctx.tensor(5.0 * np.eye(D))
        self.mu0 = ctx.zeros(D)
        self.b0 = 2.0
        self.WI = ctx.tensor(np.eye(D))
        self.nu0 = float(D)
        self.sumU = ctx.zeros(D)
        self.UUt = ctx.zeros(D, D)
        self.params = ctx.zeros(D + D * D)
        # what K1 needs of (mu, Lambda), written by the hyperprior draw (bdf_hyper_sample) so that K1 needs no pre-launch
        self.prior_pack = ctx.zeros(lib().bdf_prior_pack_doubles(D))
        self.prior_pack_valid = False
        # the random part of the hyperprior draw (Bartlett matrix + mean normals), drawn ahead of the rows (bdf_hyper_draws)
        self.draws = ctx.zeros(D * D + D)
        self.draws_sweep = None
        self.F = None
        self.numF = 0
        self.beta = ctx.zeros(D, 0)
        self.uhat = None
        self.mu_matrix = None
        self.Tinv = None
        self.lambda_beta = None
        self.cg_iters = None
        if not feat.isempty(en.F):
            if feat.feature_shape(en.F)[0] != en.count:
                raise ArgumentError(f"Entity {en.name} has {en.count} instances but its feature matrix has {feat.feature_shape(en.F)[0]} rows")
            self.F = FeatOperator(ctx, en.F, layout)
            self.numF = self.F.n
            self.beta = ctx.zeros(D, self.numF)          # numF x D column-major
            self.uhat = ctx.zeros(self.N, D)
            self.mu_matrix = ctx.zeros(self.N, D)
            self.Tinv = ctx.zeros(D, D)
            self.lambda_beta = ctx.tensor([en.lambda_beta])
            self.cg_iters = ctx.zeros(D, dtype=torch.int32)

    @property
    def sample(self):
        return self.bufs[self.cur]

    @property
    def sample_next(self):
        return self.bufs[(self.cur + 1) % 3]

    def rotate(self):
        self.cur = (self.cur + 1) % 3

    def host(self, name):
        t = getattr(self, name)
        if t is None:
            return np.zeros((0, 0))
        torch.cuda.synchronize(t.device)      # the state is written on several streams (rows, hyperprior): wait for all of them
        a = t.detach().cpu().numpy()
        if name in ("sample", "uhat", "mu_matrix") and self.layout.pos is not None:
            a = a[self.layout.pos]            # internal positions -> the reference's row order
        return a.T.copy() if a.ndim == 2 else a.copy()


class GibbsEngine:
    """Device state of a RelationData and the per-iteration steps of macau.jl:80-140."""

    def __init__(self, data, num_latent, seed=0, device=None, lambda_beta=float("nan"), compute_ff_size=6500,
                 full_lambda_u=True, tol=float("nan"), shard=None, chunks=None):
        if not (1 <= num_latent <= _lib.BDF_MAX_D):
            raise ArgumentError(f"num_latent={num_latent} must be in 1..{_lib.BDF_MAX_D}")
        self.data, self.D = data, int(num_latent)
        # The row context runs on a stream of its own that leaves a few CUs (one or two per XCD) free for the hyperprior's
        # small kernels, which otherwise wait for slots beside the chip-filling row kernel -- when the entities are small
        # enough for those kernels to be small (the reductions of a 10M-row entity want the whole chip)
        big = max((en.count for en in data.entities), default=0) * int(num_latent) * 8 > (32 << 20)
        # (the test rig that runs several ranks on ONE GPU, BDF_DIST_BACKEND=gloo, reserves none: two processes' kernels on the
        # same eight masked CUs stalled each other for seconds once in ten runs -- k_hyper_chain's last workgroup ran into its spin
        # bound; one process per GPU, the deployment, has the reserved CUs to itself)
        rig = shard is not None and shard[1] > 1 and os.environ.get("BDF_DIST_BACKEND") == "gloo"
        reserve = int(os.environ.get("BDF_RESERVE_CUS", "0" if (big or rig) else "8"))
        self.ctx = Context.rows(device, seed, reserve)
        if os.environ.get("BDF_ITEM_SIZE"):
            self.ctx.set_item_size(int(os.environ["BDF_ITEM_SIZE"]))
        if os.environ.get("BDF_PIECE_SIZE"):
            self.ctx.set_piece_size(int(os.environ["BDF_PIECE_SIZE"]))
        self.rank, self.world = (0, 1) if shard is None else shard
        self.full_lambda_u = bool(full_lambda_u)
        self.tol = float(tol)
        self.compute_ff_size = compute_ff_size
        # the whole iteration in one native call: entity side information, the relation models (alpha sampled, relation-level
        # side information: bdf_gibbs_set_relations) and the exchange between ranks included.  BDF_NO_NATIVE: enqueued step by
        # step from here instead (the same entry points, the same values)
        self.native = not os.environ.get("BDF_NO_NATIVE")
        # ---- row layouts (several ranks): degree of a row = its observations over all the entity's relations
        if chunks is None:
            chunks = int(os.environ.get("BDF_CHUNKS", "0"))
        # one chunk count for the whole model (a relation's modes share it): the chunked exchange -- the row kernel of chunk
        # c + 1 beside the all-gather of chunk c -- pays once an entity's share of rows is large
        if chunks <= 0:
            share = max(-(-en.count // self.world) for en in data.entities) if self.world > 1 else 0
            chunks = 8 if share >= 1_000_000 else (4 if share >= 200_000 else 1)      # the last chunk's exchange is what stays exposed
        self.layouts = []
        for en in data.entities:
            if self.world == 1:
                self.layouts.append(Layout(en.count))
                continue
            deg = np.zeros(en.count, dtype=np.int64)
            for r in en.relations:
                m = [e is en for e in r.entities].index(True)
                deg += np.bincount(np.asarray(r.data.ids[:, m], dtype=np.int64) - 1, minlength=en.count)
            self.layouts.append(Layout(en.count, deg, self.world, chunks))
        # ---- reset! (RelationData.jl:331-355)
        self.ent = []
        for j, en in enumerate(data.entities):
            if not np.isnan(lambda_beta):
                en.lambda_beta = float(lambda_beta)
            st = EntityState(self.ctx, en, self.D, j + 1, self.layouts[j])
            en.modes = [[e is en for e in r.entities].index(True) + 1 for r in en.relations]
            en.modes_other = [[k + 1 for k, e in enumerate(r.entities) if e is not en] for r in en.relations]
            if st.F is not None:
                en.use_FF = st.numF <= compute_ff_size
            from .relation_data import EntityModel
            en.model = EntityModel()
            en.model._dev = st
            self.ent.append(st)
        self.rel = []
        for r in data.relations:
            if len(r.entities) != len(r.data.dims):
                raise ArgumentError(f"Relation {r.name} has {len(r.entities)} entities but its data implies {r.data.size()}.")
            lays = [self.layouts[self._entity_index(e)] for e in r.entities]
            dr = DeviceRelation(self.ctx, r.data, lays if self.world > 1 else None, self.rank)
            r.model.mean_value = dr.value_mean()
            r._dev = dr
            self.rel.append(dr)
            # relation-level side information (RelationData.jl:348-353): FF path only, as in the reference
            dr.F = dr.beta = dr.linear = dr.train = None
            # several ranks: the relation's observations (COO order) in world blocks of obs_block; this rank's block is
            # [obs_lo, obs_hi): its rows of the relation's feature matrix, its observations as pairs (the squared-error sum
            # of sample_alpha and F'v of sample_beta_rel are summed over the ranks in rank order, bdf_sum_ranks)
            nn = r.data.nnz()
            dr.obs_block = -(-nn // self.world)
            dr.obs_lo = min(nn, self.rank * dr.obs_block)
            dr.obs_hi = min(nn, dr.obs_lo + dr.obs_block)
            if self.world > 1 and (not feat.isempty(r.F) or r.model.alpha_sample) and dr.obs_hi - dr.obs_lo < 1:
                raise ArgumentError(f"Relation {r.name} has fewer observations ({nn}) than a block per rank needs")
            if not feat.isempty(r.F):
                if feat.feature_shape(r.F)[0] != nn:
                    raise ArgumentError(f"Relation {r.name} has {nn} observations but its feature matrix has {feat.feature_shape(r.F)[0]} rows")
                dr.F = FeatOperator(self.ctx, r.F if self.world == 1 else _take_rows(r.F, np.arange(dr.obs_lo, dr.obs_hi)))
                if dr.F.n > compute_ff_size:
                    raise ArgumentError("conjugate gradient unimplemented for sampling relation beta")      # sampling.jl:335
                dr.beta = self.ctx.zeros(dr.F.n)
                dr.linear = self.ctx.tensor(np.full(dr.obs_block * self.world, r.model.mean_value))     # K1 reads [0, nnz)
                r.model.beta = np.zeros(dr.F.n)
            if dr.F is not None or r.model.alpha_sample:
                dr.train = self._pairs(self.ctx, r, np.asarray(r.data.ids)[dr.obs_lo:dr.obs_hi], np.asarray(r.data.values)[dr.obs_lo:dr.obs_hi])
                if dr.F is not None:         # pred(r) = udot + linear_values on the training table (sampling.jl:16-18)
                    check(lib().bdf_pairs_set_baseline(dr.train.handle, C.c_void_p(dr.linear.data_ptr() + 8 * dr.obs_lo)))
            dr.alpha_dev = self.ctx.tensor([float(r.model.alpha)])
        self._test_pairs = None
        self._train_pairs = None
        self._test_opts = None
        self.k1_spans = None      # bench.py: (tensor n x 2 of uint64 ticks as int64, [entity per used slot]) -- device-side launch spans (k1_span_begin)
        self.k1_events = None     # bench.py: list of (entity, KernelTimer) of the timed K1 launches
        self.k1_event_every = 1   # ... of every n-th sweep
        self._k1_sweep = 0
        # ---- streams: rows (main) | hyperpriors | prediction updates
        self.gibbs = None
        self.comm = None
        self._ev_pred = None
        self._ev_rows, self._ev_hyper = {}, {}
        if self.native:
            self._create_native()
        else:
            if self.world > 1:
                self.comm = make_comm(self.ctx, self.rank, self.world)
            self.ctx_h = self.ctx_p = self.ctx
            if not os.environ.get("BDF_NO_OVERLAP"):
                self.ctx_h = Context.side(self.ctx, reserved=True)
                if all(feat.isempty(r.F) for r in data.relations):
                    self.ctx_p = Context.side(self.ctx, [self.ctx_h])
        torch.cuda.synchronize(self.ctx.device)      # everything set up above (on torch's streams too) is in place

    # ---- the native iteration (bdf_gibbs) -----------------------------------------------------------------------------
    def _create_native(self):
        ents = (GibbsEntity * len(self.ent))()
        for j, (en, st) in enumerate(zip(self.data.entities, self.ent)):
            g = ents[j]
            g.N, g.n_real, g.tag, g.n_terms = st.N, st.n_real, st.tag, len(en.relations)
            if not en.relations:
                raise ArgumentError(f"Entity {en.name} takes part in no relation")
            for t, r in enumerate(en.relations):
                ri = [x is r for x in self.data.relations].index(True)
                g.terms[t].rel = self.rel[ri].handle
                g.terms[t].mode = en.modes[t] - 1
                for k, e2 in enumerate(r.entities):
                    g.terms[t].entity_of_mode[k] = self._entity_index(e2)
                g.terms[t].alpha = r.model.alpha
                g.terms[t].mean_value = r.model.mean_value
            for b in range(3):
                g.sample[b] = st.bufs[b].data_ptr()
            for name in ("mu", "Lambda", "mu0", "WI", "sumU", "UUt", "params", "prior_pack", "draws"):
                setattr(g, name, getattr(st, name).data_ptr())
            g.b0, g.nu0 = st.b0, st.nu0
            if st.F is not None:
                g.feat = st.F.handle
                for name in ("beta", "uhat", "mu_matrix", "Tinv", "lambda_beta", "cg_iters"):
                    setattr(g, name, getattr(st, name).data_ptr())
                g.use_ff, g.sample_lambda_beta, g.full_lambda_u = int(en.use_FF), int(en.lambda_beta_sample), int(self.full_lambda_u)
                g.tol, g.lb_nu, g.lb_mu = self.tol, en.nu, en.mu
        self.gibbs = C.c_void_p()
        check(lib().bdf_gibbs_create(self.ctx.handle, self.D, len(self.ent), ents, C.byref(self.gibbs)))
        h, p = C.c_void_p(), C.c_void_p()
        check(lib().bdf_gibbs_contexts(self.gibbs, C.byref(h), C.byref(p)))
        self.ctx_h = Context.wrap(h, self.ctx.device, self.ctx.seed)
        self.ctx_p = Context.wrap(p, self.ctx.device, self.ctx.seed)
        if self.world > 1 or os.environ.get("BDF_FORCE_COMM"):
            # (BDF_FORCE_COMM: a ONE-rank communicator all the same -- the exchange's kernels then run between the row launches
            # of a single-GPU run: the soak of the schedule with RCCL on the device, tools/soak_determinism.py rccl)
            self.comm = make_comm(self.ctx, self.rank, self.world)
            check(lib().bdf_gibbs_set_comm(self.gibbs, self.comm.handle))
        self._register_relations()

    def _register_relations(self):
        """(native iteration) the relations with a model of their own -- alpha sampled, relation-level side information: the
        library runs sample_alpha / sample_beta_rel / linear_values before the rows of every iteration (macau.jl:83-92)"""
        if not self.gibbs:
            return
        from ._lib import GibbsRelation
        rows = [(ri, r, self.rel[ri]) for ri, r in enumerate(self.data.relations) if r.model.alpha_sample or self.rel[ri].F is not None]
        arr = (GibbsRelation * max(len(rows), 1))()
        for k, (ri, r, dr) in enumerate(rows):
            g = arr[k]
            g.rel = dr.handle
            for m, e2 in enumerate(r.entities):
                g.entity_of_mode[m] = self._entity_index(e2)
            g.mean_value = r.model.mean_value
            g.alpha_dev = dr.alpha_dev.data_ptr()
            g.alpha_sample, g.rel_tag = int(bool(r.model.alpha_sample)), ri + 1
            g.alpha_lambda0, g.alpha_nu0, g.nnz = r.model.alpha_lambda0, r.model.alpha_nu0, r.data.nnz()
            g.train = dr.train.handle
            g.first_obs, g.obs_block = dr.obs_lo, dr.obs_block
            if dr.F is not None:
                g.feat, g.beta, g.linear, g.lambda_beta = dr.F.handle, dr.beta.data_ptr(), dr.linear.data_ptr(), r.model.lambda_beta
                if ri == 0 and getattr(dr, "F_test", None) is not None:
                    g.feat_test, g.test_baseline = dr.F_test.handle, dr.test_baseline.data_ptr()
        self._gibbs_relations = arr            # (the library copies the records; the tensors they point at live in self.rel)
        check(lib().bdf_gibbs_set_relations(self.gibbs, len(rows), C.cast(arr, C.c_void_p)))

    def warm_device(self, milliseconds=50.0):
        """set-up (native iteration): bring the device to its working state before the first iteration
        (bdf_gibbs_warm_device).  Full iterations -- rows of every entity, hyperprior chains, beta, the prediction kernel on
        the registered test pairs without running state -- with iteration numbers no real iteration uses, for `milliseconds`
        (several ranks: a fixed count, the same on every rank), then the chain's state is put back bit for bit.  Nothing of
        the chain advances.  (Row launches alone do not do it: a 20-iteration region behind 60 ms of them runs at 96-98 us
        per iteration, behind full iterations at 92-93: tools/region_idle_probe.py)  A no-op on the
        step-by-step path."""
        if self.gibbs and milliseconds > 0:
            check(lib().bdf_gibbs_warm_device(self.gibbs, float(milliseconds)))
            for j, st in enumerate(self.ent):          # the library's buffer rotation went on: follow it
                cur = C.c_int(0)
                check(lib().bdf_gibbs_current(self.gibbs, j, C.byref(cur)))
                st.cur = cur.value

    def set_alpha(self):
        """(native iteration) the relations' precisions are launch arguments held by the bdf_gibbs object: rebuild it after
        setPrecision! on an initialised model"""
        if self.gibbs:
            raise ArgumentError("change the precision before the engine is built")

    def k1_algorithmic_bytes(self, j):
        """bytes one K1 launch over all rows of entity j must move, summed over its relations:
        nnz*((4 + 8D)*(n_modes-1) + 8) + N*(8D + 8) [+ N*8D per-row prior mean] + 8D^2 + 8D"""
        en, st, D = self.data.entities[j], self.ent[j], self.D
        b = st.n_real * (8 * D + 8) + 8 * D * D + 8 * D + (st.n_real * 8 * D if st.F is not None else 0)
        for r in en.relations:
            b += r.data.nnz() * ((4 + 8 * D) * (len(r.entities) - 1) + 8)
        return b

    def k1_algorithmic_flops(self, j):
        """flops of one K1 launch over all rows of entity j by the reference's map (sampling.jl:200-212):
        nnz*(D(D+1) + 2D) [+ nnz*D*(n_modes-2) Hadamard] + N*(D^3/3 + 3D^2)"""
        en, st, D = self.data.entities[j], self.ent[j], self.D
        f = st.n_real * (D ** 3 / 3.0 + 3 * D * D)
        for r in en.relations:
            f += r.data.nnz() * (D * (D + 1) + 2 * D + D * max(len(r.entities) - 2, 0))
        return f

    def rows_dispatch(self, j):
        """how the library dispatched entity j's latest row launch (Context.rows_dispatch); None before the first iteration"""
        return self.ctx.rows_dispatch(self.ent[j].tag)

    def lowrank_rows(self, j):
        """rows of entity j the library draws with the low-rank sampler (k_rows_lr.hip) instead of the reference's map -- the
        library's own count for the latest launch (bdf_ctx_rows_dispatch); before the first iteration: the rule of
        bdf_launch_sample_rows restated from the environment (D > 16, one two-mode relation, no side information on the relation,
        rows of at most min(16, D / 2) observations (D > 32: min(32, D / 2)), at least 8,192 of them and at least half as many as the opposite entity has rows)"""
        got = self.rows_dispatch(j)
        if got is not None:
            return got["lowrank"]
        en, st, D = self.data.entities[j], self.ent[j], self.D
        lr = int(os.environ.get("BDF_LOWRANK", "-1"))
        cap = 32 if D > 32 else 16
        lr = min(cap, D // 2) if lr < 0 else min(lr, cap)
        if D <= 16 or lr == 0 or len(en.relations) != 1 or len(en.relations[0].entities) != 2:
            return 0
        r = en.relations[0]
        ri = [x is r for x in self.data.relations].index(True)
        if self.rel[ri].F is not None:
            return 0
        m = [e is en for e in r.entities].index(True)
        deg = np.bincount(np.asarray(r.data.ids[:, m], dtype=np.int64) - 1, minlength=en.count)
        cnt = int((deg <= lr).sum())
        min_rows = int(os.environ.get("BDF_LOWRANK_MIN_ROWS", "8192"))
        return cnt if (cnt >= max(min_rows, 1) and 2 * cnt >= r.data.dims[1 - m]) else 0

    # ---- helpers ----------------------------------------------------------------------------------------
    def _entity_index(self, en):
        return [e is en for e in self.data.entities].index(True)

    def _terms(self, j):
        en = self.data.entities[j]
        terms = (Term * len(en.relations))()
        for t, r in enumerate(en.relations):
            ri = [x is r for x in self.data.relations].index(True)
            terms[t].rel = self.rel[ri].handle
            terms[t].mode = en.modes[t] - 1
            terms[t].alpha = r.model.alpha
            terms[t].mean_value = r.model.mean_value
            terms[t].linear_values = self.rel[ri].linear.data_ptr() if self.rel[ri].linear is not None else None
            terms[t].alpha_dev = self.rel[ri].alpha_dev.data_ptr() if (self.native and r.model.alpha_sample) else None
            for k, e2 in enumerate(r.entities):
                terms[t].factors[k] = self.ent[self._entity_index(e2)].sample.data_ptr()
        return terms

    # ---- macau.jl:83-92: relation models (alpha, relation-level beta) -----------------------------------------------
    def update_relations(self):
        """alpha ~ sample_alpha(err) and beta = sample_beta_rel(r), linear_values = mean + F beta, for the relations that
        ask for them; runs on the main stream before the latent rows of the sweep (alpha is a host scalar of the row
        kernel's arguments: sampling it costs one device-to-host read per sweep, as the reference's host loop does)"""
        for ri, r in enumerate(self.data.relations):
            dr = self.rel[ri]
            if not (r.model.alpha_sample or dr.F is not None):
                continue
            facs = self.factors_of(r)
            comm = self.comm.handle if (self.world > 1 and self.comm is not None) else None
            if r.model.alpha_sample:
                sse = dr.train.sse(self.D, facs, r.model.mean_value)          # the pairs carry linear_values as baseline
                if comm is not None:
                    check(lib().bdf_sum_ranks(self.ctx.handle, comm, C.c_void_p(sse.data_ptr() + 8), 1))
                check(lib().bdf_sample_alpha(self.ctx.handle, r.model.alpha_lambda0, r.model.alpha_nu0, r.data.nnz(),
                                             C.c_void_p(sse.data_ptr() + 8), ri + 1, _ptr(dr.alpha_dev)))
                self.ctx.sync()
                r.model.alpha = float(dr.alpha_dev.item())
            if dr.F is not None:
                fp = (C.c_void_p * len(facs))(*[f.data_ptr() for f in facs])
                check(lib().bdf_sample_beta_rel_ranks(self.ctx.handle, comm, dr.F.handle, dr.train.handle, dr.obs_lo, self.D, fp,
                                                      r.model.mean_value, r.model.alpha, r.model.lambda_beta, ri + 1, _ptr(dr.beta),
                                                      C.c_void_p(dr.linear.data_ptr() + 8 * dr.obs_lo), None))
                if comm is not None:         # every rank's row kernels read linear_values of their own rows' observations
                    check(lib().bdf_allgather_block(self.ctx.handle, comm, _ptr(dr.linear), 8 * dr.obs_block))
                    check(lib().bdf_allgather_join(self.ctx.handle, comm))
        self.refresh_baselines()

    # ---- macau.jl:96-117: latent rows of entity j --------------------------------------------------------------
    def sample_entity(self, j):
        en, st = self.data.entities[j], self.ent[j]
        if not en.relations:
            raise ArgumentError(f"Entity {en.name} takes part in no relation")
        terms = self._terms(j)
        mu, is_matrix = st.mu, 0
        if st.F is not None:
            check(lib().bdf_uhat(self.ctx.handle, st.F.handle, self.D, _ptr(st.beta), _ptr(st.mu), _ptr(st.uhat), _ptr(st.mu_matrix)))
            mu, is_matrix = st.mu_matrix, 1
        timed = self.k1_events is not None and self._k1_sweep % self.k1_event_every == 0
        if timed:
            timer = KernelTimer()
            check(lib().bdf_ctx_time_next_rows(self.ctx.handle, timer.start, timer.stop))
        pack = st.prior_pack if (st.prior_pack_valid and not is_matrix) else None
        # written into the entity's next buffer (nothing this launch reads), which then becomes the current one; several ranks:
        # chunk after chunk, every chunk exchanged in place while the next one is sampled
        nch = st.layout.chunks
        for c in range(nch):
            check(lib().bdf_sample_rows(self.ctx.handle, self.D, st.N, len(terms), terms, _ptr(mu), is_matrix, _ptr(st.Lambda),
                                        st.tag, c, nch, _ptr(st.sample_next), _ptr(pack) if pack is not None else None))
            if self.comm is not None:
                check(lib().bdf_allgather_rows(self.ctx.handle, self.comm.handle, self.D, st.N, _ptr(st.sample_next), c, nch))
        if self.comm is not None:
            check(lib().bdf_allgather_join(self.ctx.handle, self.comm.handle))
        st.rotate()
        if timed:
            self.k1_events.append((j, timer))

    # ---- macau.jl:119-134: hyperprior of entity j ----------------------------------------------------------------
    def _hyper_nu(self, j):
        st = self.ent[j]
        return st.nu0 + (st.numF if (st.F is not None and self.full_lambda_u) else 0)

    def prepare_prior(self, j, sweep):
        """the data-independent part of update_prior(j) of this sweep; may be issued before the rows of j are sampled"""
        st = self.ent[j]
        check(lib().bdf_hyper_draws(self.ctx_h.handle, self.D, st.n_real, self._hyper_nu(j), st.tag, _ptr(st.draws)))
        st.draws_sweep = sweep

    def update_prior(self, j, sweep=None):
        en, st = self.data.entities[j], self.ent[j]
        L = lib()
        h = self.ctx_h.handle
        draws = st.draws if (sweep is not None and st.draws_sweep == sweep) else None
        if self.comm is not None and self.world > 1:      # own rows, then the ranks' partial sums added in rank order
            check(L.bdf_hyper_sums_ranks(h, self.comm.handle, self.D, st.N, st.layout.chunks, _ptr(st.sample),
                                         _ptr(st.uhat) if st.F is not None else None, _ptr(st.sumU), _ptr(st.UUt)))
        else:
            check(L.bdf_hyper_sums(h, self.D, st.N, _ptr(st.sample), _ptr(st.uhat) if st.F is not None else None,
                                   _ptr(st.sumU), _ptr(st.UUt)))
        nu, Tinv = st.nu0, st.WI
        if st.F is not None and self.full_lambda_u:
            nu += st.numF
            check(L.bdf_hyper_feature_terms(h, self.D, st.numF, _ptr(st.beta), _ptr(st.WI), _ptr(st.lambda_beta), _ptr(st.Tinv)))
            Tinv = st.Tinv
        check(L.bdf_hyper_sample(h, self.D, st.n_real, _ptr(st.sumU), _ptr(st.UUt), _ptr(st.mu0), st.b0, _ptr(Tinv),
                                 nu, st.tag, _ptr(st.mu), _ptr(st.Lambda), _ptr(st.params), _ptr(st.prior_pack),
                                 _ptr(draws) if draws is not None else None))
        st.prior_pack_valid = True

    # ---- macau.jl:138-140: beta of entity j ----------------------------------------------------------------
    def update_beta(self, j):
        en, st = self.data.entities[j], self.ent[j]
        if st.F is None:
            return
        # (several ranks: the conjugate-gradient columns are shared out over them and all-gathered, parallel_matrix.jl:488-507)
        check(lib().bdf_sample_beta_ranks(self.ctx.handle, self.comm.handle if self.comm is not None else None, st.F.handle, self.D,
                                          _ptr(st.sample), _ptr(st.mu), _ptr(st.Lambda), _ptr(st.lambda_beta), int(en.use_FF), self.tol, 0,
                                          int(en.lambda_beta_sample), en.nu, en.mu, st.tag, _ptr(st.beta), None, _ptr(st.cg_iters)))

    def sync_host_scalars(self):
        for en, st in zip(self.data.entities, self.ent):
            if st.lambda_beta is not None:
                en.lambda_beta = float(st.lambda_beta.item())
        for r, dr in zip(self.data.relations, self.rel):
            if dr.F is not None:
                r.model.beta = dr.beta.cpu().numpy().copy()
            if r.model.alpha_sample and self.native:       # (step by step the host reads it every iteration)
                r.model.alpha = float(dr.alpha_dev.item())

    # ---- one Gibbs iteration (the timed unit of bench.py) ------------------------------------------------------
    def step(self, i, phase, clamp=(), class_cut=0.0):
        """iteration i and the reporting step of macau.jl:142-184 on the test pairs (phase 0 burn-in, 1 first posterior
        sample, 2 later ones); returns the pairs' device stats"""
        test = self.test_pairs()
        if self.native:
            self.register_test(clamp, class_cut)
            self.sweep(i, phase)
            return test.stats
        self.sweep(i)
        r = self.data.relations[0]
        return test.update(self.D, self.factors_of(r), r.model.mean_value, phase, list(clamp), class_cut)

    def register_test(self, clamp=(), class_cut=0.0):
        """(native iteration) the test pairs and reporting options the library's prediction update runs with; step() does it,
        a caller that warms the device before its first step does it first so that the warm-up runs the prediction kernel too"""
        if not self.native:
            return
        test = self.test_pairs()
        opts = (tuple(clamp), float(class_cut))
        if self._test_opts != opts:
            lo, hi = (clamp[0], clamp[1]) if len(clamp) else (1.0, -1.0)
            r = self.data.relations[0]
            eom = (C.c_int32 * len(r.entities))(*[self._entity_index(e) for e in r.entities])
            check(lib().bdf_gibbs_set_test(self.gibbs, test.handle, eom, r.model.mean_value, lo, hi, class_cut, _ptr(test.stats)))
            self._test_opts = opts

    def k1_span_begin(self, n):
        """the next n row launches of the native iteration leave {first wave's start, last wave's end} (100 MHz ticks) in a device
        buffer (bdf_gibbs_span_rows: K1c launches only) -- their durations with no event packets around them"""
        t = torch.zeros(n, 64, 2, dtype=torch.int64, device=self.ctx.device)      # (64 shards per launch: wave w uses shard w % 64)
        t[:, :, 0] = -1                    # (uint64 all ones: the kernel takes an atomic min)
        torch.cuda.synchronize(self.ctx.device)
        self.k1_spans = (t, [])

    def k1_span_result(self):
        """-> [(entity, microseconds)] of the launches that recorded a span"""
        if self.k1_spans is None:
            return []
        self.sync()
        t, ents = self.k1_spans
        self.k1_spans = None
        h = t.cpu().numpy().view(np.uint64)
        out = []
        for k, j in enumerate(ents):
            used = h[k, :, 1] > 0
            if used.any():
                out.append((j, (int(h[k, used, 1].max()) - int(h[k, used, 0].min())) / 100.0))
        return out

    def sweep(self, i, predict_phase=None):
        """iteration i without reporting (native: with the prediction update of `predict_phase` on the registered test pairs)"""
        self._k1_sweep = i
        if self.native and self.k1_spans is not None:
            t, ents = self.k1_spans
            for j in range(len(self.ent)):
                if len(ents) < t.shape[0]:
                    check(lib().bdf_gibbs_span_rows(self.gibbs, j, C.c_void_p(t[len(ents)].data_ptr())))
                    ents.append(j)
        if self.native:
            timed = self.k1_events is not None and i % self.k1_event_every == 0
            if timed:
                for j in range(len(self.ent)):
                    timer = KernelTimer()
                    check(lib().bdf_gibbs_time_rows(self.gibbs, j, timer.start, timer.stop))
                    self.k1_events.append((j, timer))
            check(lib().bdf_gibbs_sweep(self.gibbs, C.c_uint32(int(i)), -1 if (predict_phase is None or self._test_opts is None) else int(predict_phase)))
            for st in self.ent:
                st.rotate()
                st.prior_pack_valid = True
            return
        main, side = self.ctx.stream, self.ctx_h.stream
        two = self.ctx_h is not self.ctx
        self.ctx.set_sweep(i)
        if two:
            self.ctx_h.set_sweep(i)
        three = self.ctx_p is not self.ctx
        if three:
            self.ctx_p.set_sweep(i)
            # The row kernels of the NEXT sweep overwrite the buffers that held the rows of sweep i-2.  The prediction
            # updates that read those were all enqueued before the previous sweep began.  The side stream waits for them
            # here, where it idles until this sweep's first row kernel ends; every row kernel of the next sweep waits for a
            # hyperprior event recorded on the side stream after this point.
            if self._ev_pred is not None:
                side.wait_event(self._ev_pred)
            self._ev_pred = torch.cuda.Event()
            self._ev_pred.record(self.ctx_p.stream)
        self.update_relations()
        for j in range(len(self.ent)):
            if two and j in self._ev_hyper:
                main.wait_event(self._ev_hyper[j])       # (mu, Lambda) of entity j from the previous iteration
            if two:
                self.prepare_prior(j, i)                 # side stream, beside the row sampling
            self.sample_entity(j)
            if two:
                ev = self._ev_rows.setdefault(j, torch.cuda.Event())
                ev.record(main)
                side.wait_event(ev)
            self.update_prior(j, i)
            if two:
                ev = self._ev_hyper.setdefault(j, torch.cuda.Event())
                ev.record(side)
        if three:                                         # prediction updates read this sweep's rows
            self.ctx_p.stream.wait_event(self._ev_rows[len(self.ent) - 1])
        for j in range(len(self.ent)):
            if self.ent[j].F is not None:
                if two:
                    main.wait_event(self._ev_hyper[j])
                self.update_beta(j)
                if two:                                  # the next hyperprior of j reads beta / lambda_beta
                    ev = self._ev_rows.setdefault(j, torch.cuda.Event())
                    ev.record(main)
                    side.wait_event(ev)

    def sync(self):
        if self.gibbs:
            check(lib().bdf_gibbs_sync(self.gibbs))
            return
        if self.ctx_p is not self.ctx:
            self.ctx_p.sync()
        if self.ctx_h is not self.ctx:
            self.ctx_h.sync()
        self.ctx.sync()

    # ---- predictions ------------------------------------------------------------------------------------------
    def factors_of(self, r):
        return [self.ent[self._entity_index(e)].sample for e in r.entities]

    def pred_all(self, r):
        """pred_all(r) (sampling.jl:91-97): udot over every cell + mean_value (bdf_predict_all: k_predict.hip)"""
        S = []
        for e in r.entities:
            st = self.ent[self._entity_index(e)]
            S.append(st.sample if st.layout.pos is None else st.sample[torch.as_tensor(st.layout.pos.astype(np.int64), device=st.sample.device)])
        dims = [int(x.shape[0]) for x in S]
        with torch.cuda.stream(self.ctx.stream):
            S = [x.contiguous() for x in S]
            out = torch.empty(dims, dtype=torch.float64, device=S[0].device)
        fp = (C.c_void_p * len(S))(*[x.data_ptr() for x in S])
        check(lib().bdf_predict_all(self.ctx.handle, len(S), (C.c_int64 * len(S))(*dims), self.D, fp, float(r.model.mean_value),
                                    C.c_void_p(out.data_ptr())))
        return out

    def _pairs(self, ctx, r, ids, values):
        """test_vec / the training table as device pairs, ids at the entities' internal positions"""
        ids = np.asarray(ids).reshape(len(values), len(r.entities))
        if self.world > 1:
            ids = np.stack([self.layouts[self._entity_index(e)].to_internal(ids[:, k]) for k, e in enumerate(r.entities)], axis=1)
        return DevicePairs(ctx, ids, values)

    def test_pairs(self, subset=None):
        """the relation's test_vec on the device (subset: the rows of test_vec this rank predicts)"""
        r = self.data.relations[0]
        if self._test_pairs is None:
            ids = r.test_vec.ids.reshape(len(r.test_vec), len(r.entities))
            vals = np.asarray(r.test_vec.values)
            if subset is not None:
                ids, vals = ids[subset], vals[subset]
            # native iteration: the pairs belong to the row context (the library updates them on its own prediction stream)
            self._test_pairs = self._pairs(self.ctx if self.native else self.ctx_p, r, ids, vals)
            if len(r.entities) == 2 and not os.environ.get("BDF_NO_PAIR_SORT"):
                # stored sorted by the mode with the fewest rows (most pairs per row): the update keeps that mode's factor
                # row in registers over a run of pairs and gathers only the other mode's (k_predict_runs); results stay in
                # the caller's order
                self._test_pairs.sort(int(np.argmin(r.data.dims)))
            dr = self.rel[0]
            if dr.F is not None:             # pred(r, probe_vec, F) = udot + F_test beta + mean_value (sampling.jl:9-14)
                if feat.isempty(r.test_F):
                    raise ArgumentError(f"Relation {r.name} has features but its test set has no feature rows (test_F)")
                dr.F_test = FeatOperator(self.ctx, r.test_F if subset is None else _take_rows(r.test_F, np.asarray(subset)))
                dr.test_baseline = self.ctx.zeros(self._test_pairs.n)
                check(lib().bdf_pairs_set_baseline(self._test_pairs.handle, _ptr(dr.test_baseline)))
                self._register_relations()          # (native iteration: the library refreshes the baseline after every relation beta)
        return self._test_pairs

    def refresh_baselines(self):
        """after the relation beta of this sweep: the test pairs' baseline mean_value + F_test beta (first relation)"""
        dr = self.rel[0]
        if dr.F is not None and self._test_pairs is not None:
            r = self.data.relations[0]
            check(lib().bdf_feat_linear(self.ctx.handle, dr.F_test.handle, _ptr(dr.beta), r.model.mean_value, _ptr(dr.test_baseline)))

    def train_pairs(self):
        r = self.data.relations[0]
        if self._train_pairs is None and self.rel[0].train is not None:
            self._train_pairs = self.rel[0].train
        if self._train_pairs is None:
            self._train_pairs = self._pairs(self.ctx_p if not self.native else self.ctx, r, r.data.ids, r.data.values)
        return self._train_pairs

    def close(self):
        if not self.ctx.handle:
            return
        try:
            self.sync()
        except Exception:
            pass
        for p in (self._test_pairs, self._train_pairs):
            if p is not None:
                p.close()
        for st in self.ent:
            if st.F is not None:
                st.F.close()
        for dr in self.rel:
            dr.close()
        if self.gibbs:
            lib().bdf_gibbs_destroy(self.gibbs)          # (and its two contexts)
            self.gibbs = None
            self.ctx_h.handle = self.ctx_p.handle = C.c_void_p()
        else:
            if self.ctx_p is not self.ctx:
                self.ctx_p.close()
            if self.ctx_h is not self.ctx:
                self.ctx_h.close()
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        self.ctx.close()


# ---- the communicator of a multi-rank run ----------------------------------------------------------------------------
class Comm:
    """bdf_comm: RCCL (one GPU per rank; the unique id travels over torch.distributed's channel), or -- several ranks on one
    GPU, the test rig BDF_DIST_BACKEND=gloo -- the library's host transport with a gloo all-gather behind it"""

    def __init__(self, ctx, rank, world):
        import torch.distributed as dist
        self.handle = C.c_void_p()
        self._cb = None
        self.transport = "RCCL (ncclAllGather in place, librccl resolved by the library)"
        if dist.get_backend() == "nccl":
            # every rank must end up on the same transport.  First the LOCAL preconditions of the library's own communicator --
            # librccl resolvable (bdf_comm_unique_id loads it and asks it for an id) -- agreed on over torch's process group
            # BEFORE any rank enters ncclCommInitRank: a rank that failed here alone would otherwise skip the collective
            # initialisation the others then block in
            err = ""
            raw0 = (C.c_char * _lib.BDF_COMM_ID_BYTES)()
            try:
                if os.environ.get("BDF_COMM_FORCE_STAGED"):
                    raise _lib.HipError("BDF_COMM_FORCE_STAGED is set")
                check(lib().bdf_comm_unique_id(raw0))
            except Exception as e:          # noqa: BLE001 -- agreed on below, then reported
                err = f"{type(e).__name__}: {e}"
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=ctx.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            pre_ok = int(ok.item()) == 1
            buf = torch.zeros(_lib.BDF_COMM_ID_BYTES, dtype=torch.uint8)
            if rank == 0 and pre_ok:
                buf = torch.frombuffer(bytearray(raw0.raw), dtype=torch.uint8).clone()
            dev = buf.to(ctx.device)
            dist.broadcast(dev, src=0)
            raw = bytes(dev.cpu().numpy().tobytes())
            # every rank must end up on the same transport: the outcome of the library's own communicator is agreed on over
            # torch's process group; if any rank could not create it, all of them exchange through torch.distributed instead
            # (the library's host transport: staged through host memory -- slower, and said so in `transport`)
            # ... then the collective initialisation itself (every rank enters it, or none does), its outcome agreed on again
            try:
                if pre_ok:
                    check(lib().bdf_comm_create(ctx.handle, rank, world, raw, C.byref(self.handle)))
                elif not err:
                    err = "another rank cannot load librccl"
            except Exception as e:          # noqa: BLE001 -- agreed on below, then reported
                err = f"{type(e).__name__}: {e}"
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=ctx.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if self.handle:
                    lib().bdf_comm_destroy(self.handle)
                    self.handle = C.c_void_p()
                self.transport = ("torch.distributed all-gather behind the library's host transport (a rank could not create the "
                                  "library's RCCL communicator" + (f": {err}" if err else "") + ")")
                print(f"[bdf] rank {rank}: {self.transport}", file=sys.stderr, flush=True)
                self._host_transport(ctx, rank, world, on_device=True)
        else:
            self.transport = "host transport with a gloo all-gather behind it (test rig: several ranks on one GPU)"
            self._host_transport(ctx, rank, world, on_device=False)
        self._enable_peer(ctx, rank, world, on_device=dist.get_backend() == "nccl")

    def _enable_peer(self, ctx, rank, world, on_device):
        """large exchanges (>= BDF_COMM_PEER_MIN_BYTES per rank, default 4 MiB) by direct all-pairs copies over IPC mappings
        (bdf_comm_enable_peer; ordered on the device by interprocess events) -- after a collective self-test: every rank pulls a
        small block from every other one and checks it; if any rank fails (no peer access, IPC refused) all of them stay on the
        communicator's own transport.  Off unless BDF_COMM_PEER=1: unmeasured on a node with several GPUs."""
        import torch.distributed as dist
        self.peer_min_bytes = None
        # (opt-in until it has run on a node with several GPUs: BDF_COMM_PEER=1)
        if world <= 1 or os.environ.get("BDF_COMM_PEER", "0") != "1":
            return
        self._peer_cb = _lib.EXCHANGE_FN(self._make_exchange(ctx, world, on_device))
        min_bytes = int(os.environ.get("BDF_COMM_PEER_MIN_BYTES", str(4 << 20)))
        err = ""
        try:
            check(lib().bdf_comm_enable_peer(self.handle, self._peer_cb, None, 0))
            probe = torch.full((world, 64), -1.0, dtype=torch.float64, device=ctx.device)
            probe[rank] = float(rank + 1)
            torch.cuda.synchronize(ctx.device)
            check(lib().bdf_comm_peer_selftest(ctx.handle, self.handle, C.c_void_p(probe.data_ptr()), 64 * 8))
            torch.cuda.synchronize(ctx.device)
            want = torch.arange(1, world + 1, dtype=torch.float64, device=ctx.device)[:, None].expand(world, 64)
            if not torch.equal(probe, want):
                err = "the self-test's blocks did not arrive"
        except Exception as e:          # noqa: BLE001 -- agreed on below
            err = f"{type(e).__name__}: {e}"
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=ctx.device if on_device else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            lib().bdf_comm_disable_peer(self.handle)
            if err:
                print(f"[bdf] rank {rank}: direct peer copies are off ({err})", file=sys.stderr, flush=True)
            return
        check(lib().bdf_comm_enable_peer(self.handle, self._peer_cb, None, min_bytes))
        self.peer_min_bytes = min_bytes
        self.transport += (f"; row exchanges of >= {min_bytes} bytes per rank by direct all-pairs peer copies over IPC mappings, "
                           f"ordered on the device by interprocess events (bdf_comm_enable_peer: one copy per xGMI link at once instead of the ring)")

    def peer_stats(self):
        n, b = C.c_int64(0), C.c_int64(0)
        check(lib().bdf_comm_peer_stats(self.handle, C.byref(n), C.byref(b)))
        return n.value, b.value

    def _make_exchange(self, ctx, world, on_device):
        import torch.distributed as dist

        def exchange(user, send, recv, nbytes):
            try:
                s = torch.frombuffer((C.c_char * nbytes).from_address(send), dtype=torch.uint8)
                r = torch.frombuffer((C.c_char * (nbytes * world)).from_address(recv), dtype=torch.uint8)
                if on_device:               # torch's own RCCL process group: through device tensors
                    rd = torch.empty(nbytes * world, dtype=torch.uint8, device=ctx.device)
                    dist.all_gather_into_tensor(rd, s.to(ctx.device))
                    r.copy_(rd.cpu())
                else:
                    dist.all_gather_into_tensor(r, s)
                return 0
            except Exception:        # noqa: BLE001 -- reported through the library's error code
                return 1
        return exchange

    def _host_transport(self, ctx, rank, world, on_device):
        self._cb = _lib.EXCHANGE_FN(self._make_exchange(ctx, world, on_device))
        check(lib().bdf_comm_create_host(ctx.handle, rank, world, self._cb, None, C.byref(self.handle)))

    def close(self):
        if self.handle:
            lib().bdf_comm_destroy(self.handle)
            self.handle = C.c_void_p()


def make_comm(ctx, rank, world):
    return Comm(ctx, rank, world)
