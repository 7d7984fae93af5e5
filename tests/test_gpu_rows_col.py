"""GPU parity of K1c (k_rows_col.hip): the latent rows of one two-mode relation at 16 < D <= 32, four rows per wave in the
column layout, against the CPU oracle (sample_user_basic, /root/reference/src/sampling.jl:200-212) on the same normals.

Every call goes through the C ABI (bdf_sample_rows with bdf_ctx_set_col_rows).  Tolerance: 1e-8 relative against the oracle
(which follows the reference's inv + chol(covar)); 1e-10 against the wave-per-row kernel (the same factorisation of the same
matrix, another order of the floating-point sums); bit-identical between shards, piece-independent launches and repeats.
"""
import numpy as np
import pytest

from test_gpu_rows import SEED, _dev_terms, _run_rows

pytestmark = pytest.mark.gpu


def _ragged(rng, dims, head, hi, coded):
    deg = rng.integers(0, hi, dims[0])
    deg[:len(head)] = head
    rows = np.repeat(np.arange(1, dims[0] + 1), deg)
    ids = np.stack([rows, rng.integers(1, dims[1] + 1, len(rows))], axis=1).astype(np.int64)
    vals = rng.integers(1, 6, len(rows)).astype(np.float64) if coded else rng.random(len(rows)) * 4 + 1
    return ids, vals, deg


@pytest.mark.parametrize("D", [17, 20, 24, 27, 30, 32])
@pytest.mark.parametrize("coded", [True, False])
def test_col_rows_against_oracle(B, O, ctx, D, coded):
    """Rows of 0 / 1 / 15 / 16 / 17 / 31 / 32 / 33 / 63 / 64 / 65 / 150 / 700 observations with pieces of at most 16: whole rows,
    rows cut into two and four pieces on the lane rows of one wave, rows that span waves (more than 64 observations: partial sums
    through the slab, the last part finishes), a row count that is no multiple of four, ratings (coded ids) and continuous values,
    shared and per-row prior means, both modes -- against the oracle at 1e-8 and against the wave-per-row kernel at 1e-10."""
    rng = np.random.default_rng(1900 + D)
    dims = [203, 90]
    ids, vals, deg = _ragged(rng, dims, [0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 150, 700, 2, 48, 49], 120, coded)
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    facs = [rng.standard_normal((d, D)) * 0.5 for d in dims]
    ft = [ctx.tensor(f) for f in facs]
    A = rng.standard_normal((D, D))
    Lam = A @ A.T / D + np.eye(D)
    alpha, mean = 1.7, float(vals.mean())
    idx = O.index_build(ids, dims)
    ctx.set_lowrank(0, 0)
    try:
        for mode0 in (0, 1):
            N = dims[mode0]
            mu = rng.standard_normal(D)
            mu_rows = rng.standard_normal((N, D))
            terms = _dev_terms(B, ctx, [(dr, mode0, alpha, mean, [None if k == mode0 else ft[k] for k in (0, 1)], None)])
            ot = O.Term(ids, vals, dims, mode0, alpha, mean, [None if k == mode0 else facs[k] for k in (0, 1)], index=idx)
            Lam_t = ctx.tensor(Lam)
            for per_row in (False, True):
                m = mu_rows if per_row else mu
                mu_t = ctx.tensor(m)
                ctx.set_sweep(6)
                got = {}
                for piece in (16, 0):
                    ctx.set_col_rows(piece)
                    out_t = ctx.zeros(N, D)
                    _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 4, out_t)
                    got[piece] = out_t.cpu().numpy()
                exp = O.sample_rows(D, N, [ot], m, Lam, SEED, 6, 4)
                assert np.isfinite(got[16]).all()
                np.testing.assert_allclose(got[16], exp, rtol=1e-8, atol=1e-9)
                np.testing.assert_allclose(got[16], got[0], rtol=1e-10, atol=1e-11)
                assert ctx.rows_unfinished() == 0
    finally:
        ctx.set_col_rows(-1)
        ctx.set_lowrank(-1, 8192)
    dr.close()


def test_col_rows_shards_repeats_and_default_piece(B, O, ctx):
    """D = 32 with the default pieces (128): a launch cut into three shards writes the same bits as one launch (how a row is cut
    depends on its own length only), a repeated launch repeats them, and rows of up to 2,000 observations (eight parts of a
    spanning row) match the oracle."""
    D = 32
    rng = np.random.default_rng(77)
    dims = [611, 300]
    ids, vals, deg = _ragged(rng, dims, [2000, 1025, 1024, 257, 256, 255, 129, 128, 127, 65, 64, 0, 0, 0, 1], 200, True)
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    facs = [rng.standard_normal((d, D)) * 0.4 for d in dims]
    ft = [ctx.tensor(f) for f in facs]
    A = rng.standard_normal((D, D))
    Lam, mu = A @ A.T / D + np.eye(D), rng.standard_normal(D)
    alpha, mean = 2.0, float(vals.mean())
    N = dims[0]
    terms = _dev_terms(B, ctx, [(dr, 0, alpha, mean, [None, ft[1]], None)])
    ot = O.Term(ids, vals, dims, 0, alpha, mean, [None, facs[1]])
    Lam_t, mu_t = ctx.tensor(Lam), ctx.tensor(mu)
    ctx.set_lowrank(0, 0)
    ctx.set_col_rows(128)
    try:
        ctx.set_sweep(11)
        one = ctx.zeros(N, D)
        _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 3, one)
        again = ctx.zeros(N, D)
        _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 3, again)
        parts = ctx.zeros(N, D)
        for s in range(3):
            _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 3, parts, shard=s, n_shards=3)
        exp = O.sample_rows(D, N, [ot], mu, Lam, SEED, 11, 3)
        np.testing.assert_allclose(one.cpu().numpy(), exp, rtol=1e-8, atol=1e-9)
        assert np.array_equal(one.cpu().numpy(), again.cpu().numpy())
        assert np.array_equal(one.cpu().numpy(), parts.cpu().numpy())
        assert ctx.rows_unfinished() == 0
    finally:
        ctx.set_col_rows(-1)
        ctx.set_lowrank(-1, 8192)
    dr.close()


def test_col_rows_piece_size_follows_the_entity_and_not_the_shard(B, O, ctx):
    """An entity of many observations takes larger pieces (bdf_launch_sample_rows: its observation count over 8 x 2,048 nominal slots,
    between 128 and 2,048) -- here 600 rows of 4,200 observations at D = 30, 2.5 M in all: T = 192, every row spans waves in six parts.
    The cut comes from the WHOLE entity's count: three shards write the bits of one launch, and the library reports the dispatch
    (bdf_ctx_rows_dispatch).  Against the oracle at 1e-8."""
    D = 30
    rng = np.random.default_rng(4200)
    dims = [600, 9000]
    per = 4200
    rows = np.repeat(np.arange(1, dims[0] + 1), per)
    cols = np.concatenate([np.sort(rng.choice(dims[1], per, replace=False)) + 1 for _ in range(dims[0])])
    ids = np.stack([rows, cols], axis=1).astype(np.int64)
    vals = rng.random(len(rows))
    dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
    facs = [rng.standard_normal((d, D)) * 0.3 for d in dims]
    ft = [ctx.tensor(f) for f in facs]
    A = rng.standard_normal((D, D))
    Lam, mu = A @ A.T / D + np.eye(D), rng.standard_normal(D)
    alpha, mean = 1.3, float(vals.mean())
    N = dims[0]
    terms = _dev_terms(B, ctx, [(dr, 0, alpha, mean, [None, ft[1]], None)])
    ot = O.Term(ids, vals, dims, 0, alpha, mean, [None, facs[1]])
    Lam_t, mu_t = ctx.tensor(Lam), ctx.tensor(mu)
    ctx.set_lowrank(0, 0)
    try:
        ctx.set_sweep(5)
        one = ctx.zeros(N, D)
        _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 9, one)
        d = ctx.rows_dispatch(9)
        assert d["col"] == N and d["k1"] == 0
        T = (len(rows) // (2048 * 8) + 63) // 64 * 64
        assert T == 192 and d["col_waves"] == N * -(-per // (4 * T))
        parts = ctx.zeros(N, D)
        for s in range(3):
            _run_rows(B, ctx, D, N, terms, mu_t, Lam_t, 9, parts, shard=s, n_shards=3)
        exp = O.sample_rows(D, N, [ot], mu, Lam, SEED, 5, 9)
        np.testing.assert_allclose(one.cpu().numpy(), exp, rtol=1e-8, atol=1e-9)
        assert np.array_equal(one.cpu().numpy(), parts.cpu().numpy())
        assert ctx.rows_unfinished() == 0
    finally:
        ctx.set_lowrank(-1, 8192)
    dr.close()
