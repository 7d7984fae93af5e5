"""The engine's own test-pair object (engine.test_pairs(): it lives on the engine's PREDICTION stream and is stored sorted
by the smaller mode) against the oracle's pred(r, probe_vec) (src/sampling.jl:9-27): the path smoke() takes.  The
round-1 predict tests all used one single-stream Context and could not see a cross-stream ordering bug in
DevicePairs.predict.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["native", "stepwise"])
def mode(request, monkeypatch):
    """the native iteration (bdf_gibbs_sweep; the pairs belong to the row context) and the step-by-step one (the pairs live on
    the engine's prediction stream: the configuration of the round-1 race)"""
    if request.param == "stepwise":
        monkeypatch.setenv("BDF_NO_NATIVE", "1")
    return request.param


def _smoke_like(B, N1, N2, D, nnz, ntest, seed):
    rng = np.random.default_rng(seed)
    ids = np.stack([rng.integers(1, N1 + 1, nnz), rng.integers(1, N2 + 1, nnz)], axis=1)
    U, V = rng.standard_normal((N1, 3)), rng.standard_normal((N2, 3))
    vals = np.sum(U[ids[:, 0] - 1] * V[ids[:, 1] - 1], axis=1) + 0.1 * rng.standard_normal(nnz)
    rel = B.Relation({"u": ids[:, 0], "v": ids[:, 1], "y": vals}, "r", [B.Entity("u"), B.Entity("v")], dims=[N1, N2])
    B.assignToTest(rel, np.arange(1, ntest + 1))
    B.setPrecision(rel, 2.0)
    return B.RelationData(rel), rel


@pytest.mark.parametrize("N1,N2,D", [(300, 200, 16), (200, 300, 16), (257, 123, 32), (90, 140, 7)])
def test_engine_test_pairs_predict_matches_oracle(B, O, mode, N1, N2, D):
    """two sweeps on a multi-stream engine, then predict() on the engine's pairs, repeatedly and without any host
    synchronisation in between: every call must see the rows of the sweep enqueued before it"""
    import torch
    rd, rel = _smoke_like(B, N1, N2, D, 6000, 500, seed=N1 + D)
    eng = B.GibbsEngine(rd, D, seed=42)
    assert eng.ctx_p is not eng.ctx and eng.ctx_p.stream != eng.ctx.stream        # rows and predictions on different streams
    tp = eng.test_pairs()
    assert eng.native == (mode == "native") and (tp.ctx is eng.ctx_p) == (mode == "stepwise")
    for it in range(1, 6):
        eng.sweep(it)
        with torch.cuda.stream(eng.ctx.stream):
            pred = tp.predict(D, eng.factors_of(rel), rel.model.mean_value)
            got = pred.cpu().numpy()                                                # no eng.sync() before the read-back
        S = [eng.ent[j].host("sample").T for j in (0, 1)]                           # (N_j, D) of this sweep
        exp = O.predict(rel.test_vec.ids, S, rel.model.mean_value)
        np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-12)
    eng.close()


def test_engine_test_pairs_predict_movielens_d32(B, O):
    from bdf_amd import datasets
    rd, source = datasets.movielens_relation_data(B)
    rel = rd.relations[0]
    D = 32
    eng = B.GibbsEngine(rd, D, seed=5)
    tp = eng.test_pairs()
    for it in range(1, 4):
        eng.sweep(it)
    got = tp.predict(D, eng.factors_of(rel), rel.model.mean_value).cpu().numpy()
    S = [eng.ent[j].host("sample").T for j in (0, 1)]
    ids = np.asarray(rel.test_vec.ids).reshape(-1, 2)
    exp = np.einsum("nd,nd->n", S[0][ids[:, 0] - 1], S[1][ids[:, 1] - 1]) + rel.model.mean_value
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-12)
    sub = np.arange(0, len(ids), 997)
    np.testing.assert_allclose(got[sub], O.predict(ids[sub], S, rel.model.mean_value), rtol=1e-12, atol=1e-12)
    eng.close()
