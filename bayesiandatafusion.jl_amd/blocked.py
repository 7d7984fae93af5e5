"""Block / sample_users_blocked (src/sampling.jl:236-249): the users of a block all observed the same items, so the
reference shares one covariance between them:
    covar = inv(Lambda_u + alpha MM MM'),  mu = covar (alpha MM Yma + Lambda_u mu_u),  chol(covar)' z + mu.
bdf_sample_block (csrc/k_block.hip) does the same on the device: the block's precision matrix is accumulated and factored
ONCE, every user then costs its right-hand side and two triangular solves.  shared=False takes the row kernel instead (the
block as a dense relation: one factorisation per user) -- same values, kept as the cross-check.
(macau_blocked.jl, the only would-be caller, is an empty stub in the reference.)"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError, DimensionMismatch, Term, check, lib
from .indexed_df import IndexedDF


class Block:
    """type Block (sampling.jl:236-240): ux ids of the latent variables (1-based), vx ids of the other side (1-based),
    Yma the values without their mean, length(vx) x length(ux)."""

    def __init__(self, ux, vx, Yma):
        self.ux = np.asarray(ux, dtype=np.int64)
        self.vx = np.asarray(vx, dtype=np.int64)
        self.Yma = np.asarray(Yma, dtype=np.float64)
        if self.Yma.shape != (len(self.vx), len(self.ux)):
            raise DimensionMismatch(f"Yma is {self.Yma.shape}, expected ({len(self.vx)}, {len(self.ux)})")


def sample_users_blocked(block, sample_mt, alpha, mu_u, Lambda_u, ctx=None, entity_tag=1, shared=True):
    """sample_users_blocked(block, sample_mt, alpha, mu_u, Lambda_u) (sampling.jl:242-249) -> D x length(block.ux).
    sample_mt: D x M sample of the other side.  Column u of the result uses the normals of stream (row, entity_tag,
    u - 1) of the context's current sweep (column index within the block, not block.ux[u])."""
    from .engine import Context, DeviceRelation, _ptr
    sample_mt = np.asarray(sample_mt, dtype=np.float64)
    D, M = sample_mt.shape
    mu_u, Lambda_u = np.asarray(mu_u, dtype=np.float64), np.asarray(Lambda_u, dtype=np.float64)
    if mu_u.shape != (D,) or Lambda_u.shape != (D, D):
        raise DimensionMismatch(f"mu_u {mu_u.shape} / Lambda_u {Lambda_u.shape} do not match num_latent={D}")
    if len(block.vx) and (block.vx.min() < 1 or block.vx.max() > M):
        raise ArgumentError(f"block.vx must be in 1..{M}")
    own = ctx is None
    ctx = Context() if own else ctx
    try:
        nu, nv = len(block.ux), len(block.vx)
        if nu == 0:
            return np.zeros((D, 0))
        if shared:
            import torch
            fac = ctx.tensor(sample_mt.T)                        # M x D, a gathered column of sample_mt is contiguous
            vx = ctx.tensor(block.vx - 1, dtype=torch.int32)
            Y = ctx.tensor(np.asfortranarray(block.Yma).T)       # (nu, nv) C order == nv x nu column-major
            mu_t, Lam_t = ctx.tensor(mu_u), ctx.tensor(Lambda_u)
            out = ctx.zeros(nu, D)
            check(lib().bdf_sample_block(ctx.handle, D, nu, nv, _ptr(vx), _ptr(Y), _ptr(fac), float(alpha), _ptr(mu_t), _ptr(Lam_t),
                                         int(entity_tag), _ptr(out)))
            ctx.sync()
            return out.cpu().numpy().T.copy()
        ids = np.stack([np.repeat(np.arange(1, nu + 1), nv), np.tile(block.vx, nu)], axis=1)
        vals = block.Yma.T.reshape(-1)                       # user-major: all items of user 1, then user 2, ...
        dr = DeviceRelation(ctx, IndexedDF((ids, vals), [nu, M]))
        fac = ctx.tensor(sample_mt.T)                        # M x D, a gathered column of sample_mt is contiguous
        terms = (Term * 1)()
        terms[0].rel = dr.handle
        terms[0].mode = 0
        terms[0].alpha = float(alpha)
        terms[0].mean_value = 0.0                            # Yma is "Y values w/o mean"
        terms[0].linear_values = None
        terms[0].factors[1] = fac.data_ptr()
        mu_t, Lam_t = ctx.tensor(mu_u), ctx.tensor(Lambda_u)
        out = ctx.zeros(nu, D)
        check(lib().bdf_sample_rows(ctx.handle, D, nu, 1, terms, _ptr(mu_t), 0, _ptr(Lam_t), int(entity_tag), 0, 1, _ptr(out), None))
        ctx.sync()
        res = out.cpu().numpy().T.copy()
        dr.close()
        return res
    finally:
        if own:
            ctx.close()
