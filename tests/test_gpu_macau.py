"""End-to-end macau() on the GPU: the reference's own test scenarios (test/basic.jl, tensor.jl, lambda_sampling.jl,
custom_rd.jl, beta_saving.jl, heavy_copyto.jl:72-91), a step-by-step comparison of whole Gibbs iterations with the CPU
oracle, determinism, and the published MovieLens quality figure (docs/index.md:83).
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sprand(m, n, density, seed):
    import scipy.sparse as sp
    return sp.random(m, n, density=density, random_state=seed, format="csc", data_rvs=np.random.default_rng(seed).random)


def test_bpmf_smoke_and_identities(B):
    """test/basic.jl:97-123, 164-174"""
    Y = _sprand(15, 10, 0.3, 1)
    rd = B.RelationData(Y, class_cut=0.5)
    B.assignToTest(rd.relations[0], 2, rng=np.random.default_rng(0))
    assert B.numTest(rd.relations[0]) == 2 and len(rd.relations[0].test_label) == 2
    result = B.macau(rd, burnin=10, psamples=10, verbose=False)
    assert result["predictions"].shape[0] == 2
    assert len(result["predictions"]["stdev"]) == 2 and np.all(result["predictions"]["stdev"] >= 0)
    assert result["latent_multi_threading"] is True
    for key in ("num_latent", "burnin", "psamples", "lambda_beta", "RMSE", "accuracy", "ROC", "train_counts"):
        assert key in result
    Yhat = B.pred_all(rd.relations[0])
    assert Yhat.shape == (15, 10)
    s1, s2 = rd.entities[0].model.sample, rd.entities[1].model.sample
    assert s1.shape == (10, 15) and s2.shape == (10, 10)
    assert math.isclose(Yhat[1, 2], s1[:, 1] @ s2[:, 2] + rd.relations[0].model.mean_value, rel_tol=1e-12)
    # pred for the training set (basic.jl:169-174)
    yt = B.pred(rd.relations[0])
    assert len(yt) == B.numData(rd.relations[0])
    row, col = rd.relations[0].data.ids[0]
    y1 = np.sum(s1[:, row - 1] * s2[:, col - 1]) + B.valueMean(rd.relations[0].data)
    assert math.isclose(y1, yt[0], rel_tol=1e-12)
    # rmse_train, custom function
    r1 = B.macau(rd, burnin=1, psamples=2, verbose=False, rmse_train=True)
    assert r1["RMSE_train"] >= 0
    r2 = B.macau(rd, burnin=5, psamples=6, verbose=False, f=lambda a: len(a.entities))
    assert r2["f_output"] == [2] * 6


@pytest.mark.parametrize("output_type", ["binary", "csv"])
def test_sample_dumps(B, tmp_path, output_type):
    """test/basic.jl:126-161"""
    Y = _sprand(15, 10, 0.3, 2)
    rd = B.RelationData(Y, class_cut=0.5, entity1="e1", entity2="e2")
    prefix = str(tmp_path / "macau-runtest")
    B.macau(rd, burnin=1, psamples=10, verbose=False, num_latent=5, output=prefix, output_type=output_type)
    ext = "binary" if output_type == "binary" else "csv"
    for en in ("e1", "e2"):
        for k in ("01", "02", "10"):
            assert os.path.isfile(f"{prefix}-{en}-{k}.{ext}")
    rd_s = (B.read_binary_float32(f"{prefix}-e1-01.binary") if output_type == "binary"
            else np.loadtxt(f"{prefix}-e1-01.csv", delimiter=","))
    assert rd_s.shape == (5, 15)
    last = (B.read_binary_float32(f"{prefix}-e1-10.binary") if output_type == "binary"
            else np.loadtxt(f"{prefix}-e1-10.csv", delimiter=","))
    np.testing.assert_allclose(last, rd.entities[0].model.sample.astype(np.float32), rtol=1e-6)


def test_tensor_relation(B):
    """test/tensor.jl: 3-mode relation from a table, entity names from the columns, pred_all identity"""
    rng = np.random.default_rng(3)
    A, Bm, Cm = rng.standard_normal((15, 2)), rng.standard_normal((20, 2)), rng.standard_normal((2, 2))
    rows = [(i + 1, j + 1, k + 1, float(np.sum(A[i] * Bm[j] * Cm[k]))) for i in range(15) for j in range(20) for k in range(2)]
    t = np.array(rows)
    df = {"A": t[:, 0].astype(int), "B": t[:, 1].astype(int), "C": t[:, 2].astype(int), "v": t[:, 3]}
    rd = B.RelationData(df)
    assert [e.name for e in rd.entities] == ["A", "B", "C"]
    B.assignToTest(rd.relations[0], 10, rng=np.random.default_rng(1))
    result = B.macau(rd, burnin=50, psamples=10, num_latent=2, verbose=False)
    assert result["latent_multi_threading"] is True
    Yhat = B.pred_all(rd.relations[0])
    assert Yhat.shape == (15, 20, 2)
    y = rd.entities[0].model.sample[:, 3] * rd.entities[1].model.sample[:, 1] * rd.entities[2].model.sample[:, 0]
    assert math.isclose(Yhat[3, 1, 0], y.sum() + rd.relations[0].model.mean_value, rel_tol=1e-12)
    assert result["RMSE"] < 0.5          # the planted rank-2 model is recovered


def test_entity_features_ff_and_cg(B):
    """test/lambda_sampling.jl, test/custom_rd.jl:7-31, test/heavy_copyto.jl:72-77"""
    rng = np.random.default_rng(4)
    A, Bm = rng.standard_normal((30, 2)), rng.standard_normal((40, 2))
    ids = np.array([(i + 1, j + 1) for i in range(30) for j in range(40)])
    vals = np.array([A[i - 1] @ Bm[j - 1] for i, j in ids])
    rd = B.RelationData({"A": ids[:, 0], "B": ids[:, 1], "v": vals})
    rd.entities[0].F = rng.standard_normal((30, 2))
    rd.entities[0].lambda_beta_sample = True
    B.assignToTest(rd.relations[0], 10, rng=np.random.default_rng(2))
    res = B.macau(rd, burnin=50, psamples=10, num_latent=2, verbose=False)
    assert rd.entities[0].use_FF is True and rd.entities[0].lambda_beta > 0
    assert rd.entities[0].model.beta.shape == (2, 2)
    assert res["RMSE"] < 0.6
    # custom RelationData with features on both entities (custom_rd.jl)
    genes, pheno = B.Entity("genes"), B.Entity("pheno")
    genes.F = rng.random((100, 5)); genes.lambda_beta = 3.0
    pheno.F = rng.random((50, 8))
    data = {"gene": rng.integers(1, 101, 1050), "pheno": rng.integers(1, 51, 1050), "value": rng.random(1050)}
    r = B.Relation(data, "HPO", [genes, pheno], class_cut=0.5, dims=[100, 50])
    rd2 = B.RelationData()
    B.addRelation(rd2, r)
    assert r.class_cut == 0.5 and genes.count == 100 and pheno.count == 50
    assert len(genes.relations) == 1 and len(pheno.relations) == 1 and len(rd2.relations) == 1 and len(rd2.entities) == 2
    B.macau(rd2, burnin=10, psamples=10, verbose=False)
    # CG path (compute_ff_size = 0) on a sparse F (heavy_copyto.jl:72-77)
    Y = _sprand(20, 10, 0.4, 5)
    featm = _sprand(20, 5, 0.5, 6)
    rd3 = B.RelationData(Y, class_cut=0.5, feat1=featm.tocsr())
    B.assignToTest(rd3.relations[0], 2, rng=np.random.default_rng(3))
    B.macau(rd3, burnin=2, psamples=2, verbose=False, compute_ff_size=0, num_latent=5)
    assert rd3.entities[0].use_FF is False
    assert rd3.entities[0].model.beta.shape == (5, 5)


def test_beta_saving(B, tmp_path):
    """test/beta_saving.jl"""
    rng = np.random.default_rng(6)
    A, Bm = rng.standard_normal((20, 2)), rng.standard_normal((30, 2))
    ids = np.array([(i + 1, j + 1) for i in range(20) for j in range(30)])
    vals = np.array([A[i - 1] @ Bm[j - 1] for i, j in ids])
    rd = B.RelationData({"A": ids[:, 0], "B": ids[:, 1], "v": vals})
    rd.entities[0].F = rng.standard_normal((20, 3))
    B.assignToTest(rd.relations[0], 10, rng=np.random.default_rng(7))
    prefix = str(tmp_path / "macau-betasaving")
    B.macau(rd, burnin=5, psamples=10, num_latent=2, verbose=False, output_beta=True, output=prefix, output_type="binary")
    for k in ("01", "02"):
        assert os.path.isfile(f"{prefix}-A-{k}.beta.binary")
    b1 = B.read_binary_float32(f"{prefix}-A-01.beta.binary")
    assert b1.shape == (3, 2)
    b10 = B.read_binary_float32(f"{prefix}-A-10.beta.binary")
    np.testing.assert_allclose(b10, rd.entities[0].model.beta.astype(np.float32), rtol=1e-6)


def _oracle_macau(O, rd, D, seed, iters, feats, use_ff=True, lambda_beta0=1.0):
    """macau.jl:80-140 on the CPU oracle for a two-entity relation (optional dense features on entity 0)"""
    r = rd.relations[0]
    N = list(r.data.dims)
    S = [np.zeros((N[0], D)), np.zeros((N[1], D))]
    mu = [np.zeros(D), np.zeros(D)]
    Lam = [5.0 * np.eye(D), 5.0 * np.eye(D)]
    F = feats
    beta = np.zeros((F.shape[1], D)) if F is not None else None
    lb = lambda_beta0                 # Entity default (RelationData.jl:60); the GPU run has since overwritten the field
    mean = r.data.valueMean()
    for it in range(1, iters + 1):
        for j in (0, 1):
            t = O.Term(r.data.ids, r.data.values, N, j, r.model.alpha, mean, [None if k == j else S[k] for k in (0, 1)])
            if j == 0 and F is not None:
                uhat = F @ beta
                S[j] = O.sample_rows(D, N[j], [t], mu[j] + uhat, Lam[j], seed, it, j + 1)
                U, nu, Tinv = S[j] - uhat, D + F.shape[1], np.eye(D) + beta.T @ beta * lb
            else:
                S[j] = O.sample_rows(D, N[j], [t], mu[j], Lam[j], seed, it, j + 1)
                U, nu, Tinv = S[j], float(D), np.eye(D)
            mu_N, beta_N, T_N, nu_N = O.hyper_params(U, np.zeros(D), 2.0, Tinv, nu)
            mu[j], Lam[j] = O.hyper_draw(mu_N, beta_N, T_N, nu_N, seed, it, j + 1)
        if F is not None:
            beta, _, _ = O.sample_beta(O.Feat.from_dense(F), S[0], mu[0], Lam[0], lb, use_ff, None, seed, it, 1)
            lb = O.sample_lambda_beta(beta, Lam[0], 1e-3, 1.0, seed, it, 1)
    return S, mu, Lam, beta, lb


@pytest.mark.parametrize("with_feat,use_ff", [(False, True), (True, True), (True, False)])
def test_whole_iterations_match_oracle(B, O, with_feat, use_ff):
    rng = np.random.default_rng(8)
    N1, N2, D, nnz = 60, 45, 6, 900
    ids = np.stack([rng.integers(1, N1 + 1, nnz), rng.integers(1, N2 + 1, nnz)], axis=1)
    vals = rng.standard_normal(nnz)
    F = rng.standard_normal((N1, 4)) if with_feat else None
    e1, e2 = B.Entity("u", F=F), B.Entity("v")
    rel = B.Relation({"u": ids[:, 0], "v": ids[:, 1], "y": vals}, "r", [e1, e2], dims=[N1, N2])
    B.setPrecision(rel, 2.0)
    rd = B.RelationData(rel)
    B.macau(rd, burnin=3, psamples=0, num_latent=D, verbose=False, seed=77, compute_ff_size=6500 if use_ff else 0)
    S, mu, Lam, beta, lb = _oracle_macau(O, rd, D, 77, 3, F, use_ff)
    for j in (0, 1):
        np.testing.assert_allclose(rd.entities[j].model.sample.T, S[j], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rd.entities[j].model.mu, mu[j], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rd.entities[j].model.Lambda, Lam[j], rtol=1e-6, atol=1e-7)
    if with_feat:
        np.testing.assert_allclose(rd.entities[0].model.beta, beta, rtol=1e-5, atol=1e-7)
        assert math.isclose(rd.entities[0].lambda_beta, lb, rel_tol=1e-6)


def test_hyper_reference_mean_map_matches_oracle(B, O, tmp_path):
    """BDF_HYPER_MEAN=reference (a process-wide switch: a fresh process, tests/env_probe.py): the reference's own map from the mean
    normals to mu, chol(inv(Lambda) / beta_N)' z (/root/reference/src/normal_wishart.jl:38-42), on the device -- two whole native
    iterations at D = 32 (K1c rows, the one-launch hyperprior chain) against the oracle's chain with mean_map="reference"; and the
    values DIFFER from the default map's (the same law, another function of z)."""
    import os, subprocess, sys
    from env_probe import macau_problem
    seed, N1, N2, nnz, D, iters = 31, 140, 90, 5000, 32, 2
    out = str(tmp_path / "ref.npz")
    env = dict(os.environ)
    env.update({"BDF_HYPER_MEAN": "reference", "BDF_LOWRANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_probe.py"), "macau", out] +
                       [str(x) for x in (seed, N1, N2, nnz, D, iters)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    ids, vals = macau_problem(seed, N1, N2, nnz)
    N = [N1, N2]
    mean = float(vals.mean())
    res = {}
    for mm in ("reference", "factor"):
        S = [np.zeros((N1, D)), np.zeros((N2, D))]
        mu = [np.zeros(D), np.zeros(D)]
        Lam = [5.0 * np.eye(D), 5.0 * np.eye(D)]
        for it in range(1, iters + 1):
            for j in (0, 1):
                t = O.Term(ids, vals, N, j, 2.0, mean, [None if k == j else S[k] for k in (0, 1)])
                S[j] = O.sample_rows(D, N[j], [t], mu[j], Lam[j], 77, it, j + 1)
                mu_N, beta_N, T_N, nu_N = O.hyper_params(S[j], np.zeros(D), 2.0, np.eye(D), float(D))
                mu[j], Lam[j] = O.hyper_draw(mu_N, beta_N, T_N, nu_N, 77, it, j + 1, mean_map=mm)
        res[mm] = (S, mu, Lam)
    S, mu, Lam = res["reference"]
    for j in (0, 1):
        np.testing.assert_allclose(got[f"S{j}"], S[j], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got[f"mu{j}"], mu[j], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got[f"Lam{j}"], Lam[j], rtol=1e-6, atol=1e-7)
    assert not np.allclose(got["mu0"], res["factor"][1][0], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("D", [24, 32])
def test_whole_iterations_col_rows_small_pieces_match_oracle(B, O, monkeypatch, D):
    """the native iteration with K1c's pieces cut small (BDF_K1_COL=8: rows on two and four lane rows, rows that span waves, the
    kernel polling for the hyperprior's draw, per-row prior means): three whole iterations with entity side information against
    the oracle"""
    monkeypatch.setenv("BDF_K1_COL", "8")
    monkeypatch.setenv("BDF_LOWRANK", "0")
    rng = np.random.default_rng(80 + D)
    N1, N2, nnz = 150, 70, 4000
    ids = np.stack([rng.integers(1, N1 + 1, nnz), rng.integers(1, N2 + 1, nnz)], axis=1)
    vals = rng.standard_normal(nnz)
    F = rng.standard_normal((N1, 4))
    e1, e2 = B.Entity("u", F=F), B.Entity("v")
    rel = B.Relation({"u": ids[:, 0], "v": ids[:, 1], "y": vals}, "r", [e1, e2], dims=[N1, N2])
    B.setPrecision(rel, 2.0)
    rd = B.RelationData(rel)
    B.macau(rd, burnin=3, psamples=0, num_latent=D, verbose=False, seed=77)
    S, mu, Lam, beta, lb = _oracle_macau(O, rd, D, 77, 3, F, True)
    for j in (0, 1):
        np.testing.assert_allclose(rd.entities[j].model.sample.T, S[j], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rd.entities[j].model.mu, mu[j], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rd.entities[j].model.Lambda, Lam[j], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rd.entities[0].model.beta, beta, rtol=1e-5, atol=1e-7)


def test_determinism_and_stream_overlap(B, monkeypatch):
    """same seed -> bit-identical chain; the native three-stream iteration, the step-by-step one and the single-stream one
    give the same values"""
    Y = _sprand(40, 30, 0.3, 9)

    def run(seed):
        rd = B.RelationData(Y, class_cut=0.5)
        B.assignToTest(rd.relations[0], np.arange(1, 41))
        B.macau(rd, burnin=4, psamples=4, num_latent=8, verbose=False, seed=seed)
        return rd.entities[0].model.sample.copy(), rd.entities[1].model.Lambda.copy()

    a1, l1 = run(5)
    a2, l2 = run(5)
    assert np.array_equal(a1, a2) and np.array_equal(l1, l2)
    a3, _ = run(6)
    assert not np.array_equal(a1, a3)
    monkeypatch.setenv("BDF_NO_POLL", "1")              # row kernels wait for the hyperprior's event instead of polling its flag
    a7, l7 = run(5)
    assert np.array_equal(a1, a7) and np.array_equal(l1, l7)
    monkeypatch.delenv("BDF_NO_POLL")
    monkeypatch.setenv("BDF_RESERVE_CUS", "0")          # no CUs set aside for the hyperprior stream (events, whole chip)
    a8, l8 = run(5)
    assert np.array_equal(a1, a8) and np.array_equal(l1, l8)
    monkeypatch.delenv("BDF_RESERVE_CUS")
    monkeypatch.setenv("BDF_NO_NATIVE", "1")            # the iteration enqueued step by step from Python instead of bdf_gibbs_sweep
    a5, l5 = run(5)
    assert np.array_equal(a1, a5) and np.array_equal(l1, l5)
    monkeypatch.setenv("BDF_NO_OVERLAP", "1")           # ... and on one stream
    a4, l4 = run(5)
    assert np.array_equal(a1, a4) and np.array_equal(l1, l4)


def test_warm_device_and_repeated_sweep_numbers(B, monkeypatch):
    """(i) bdf_gibbs_warm_device (full iterations whose results are discarded, the chain's state put back; set-up of the bench)
    does not advance the chain; (ii) the hand-over between a draw and the next row launch is keyed on a private epoch, not on the caller's
    sweep number: a number that repeats (same random streams, evolving state) gives the same chain whether the row kernels
    poll for the draw or the row stream waits for its event"""
    from bdf_amd import datasets
    from bdf_amd.engine import GibbsEngine

    def run(warm_ms, numbers):
        rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
        eng = GibbsEngine(rd, 32, seed=11)
        assert eng.native
        if warm_ms:
            eng.warm_device(warm_ms)
        for i in numbers:
            eng.sweep(i)
            if warm_ms and i == numbers[1]:
                eng.warm_device(5.0)                  # in the middle of a chain too
        eng.sync()
        out = [e.model.sample.copy() for e in rd.entities] + [rd.entities[0].model.Lambda.copy()]
        eng.close()
        return out

    same = lambda a, b: all(np.array_equal(x, y) for x, y in zip(a, b))
    base = run(0.0, [1, 2, 3, 4])
    assert same(base, run(15.0, [1, 2, 3, 4]))
    rep = run(0.0, [7, 7, 7, 3, 3, 7])
    assert not same(base, rep)
    monkeypatch.setenv("BDF_NO_POLL", "1")
    assert same(rep, run(0.0, [7, 7, 7, 3, 3, 7]))


def test_warm_device_keeps_a_chain_with_side_information(B):
    """bdf_gibbs_warm_device with entity side information (beta, uhat, per-row prior means, lambda_beta in the state it puts
    back): a chain with a warm-up before its first iteration and one in the middle equals, bit for bit, the chain without"""
    from bdf_amd.engine import GibbsEngine
    rng = np.random.default_rng(8)
    N1, N2, nnz, numF, D = 260, 90, 5000, 30, 16
    ids = np.stack([rng.integers(1, N1 + 1, nnz), rng.integers(1, N2 + 1, nnz)], axis=1)
    vals = np.clip(np.round(3.5 + rng.standard_normal(nnz)), 1, 5)
    F = rng.standard_normal((N1, numF))

    def run(warm, ff):
        rel = B.Relation({"u": ids[:, 0], "v": ids[:, 1], "y": vals}, "r", [B.Entity("u", F=F), B.Entity("v")], dims=[N1, N2])
        B.setPrecision(rel, 1.5)
        B.assignToTest(rel, 300, rng=np.random.default_rng(2))
        rd = B.RelationData(rel)
        eng = GibbsEngine(rd, D, seed=6, compute_ff_size=6500 if ff else 0)
        assert eng.native
        eng.register_test([1.0, 5.0], rel.class_cut)
        if warm:
            eng.warm_device(8.0)
        for i in range(1, 7):
            eng.step(i, 0 if i < 4 else (1 if i == 4 else 2), [1.0, 5.0], rel.class_cut)
            if warm and i == 4:
                eng.warm_device(5.0)
        eng.sync()
        en = rd.entities[0]
        out = [en.model.sample.copy(), en.model.beta.copy(), rd.entities[1].model.sample.copy(), eng.test_pairs().stats.cpu().numpy().copy(),
               np.concatenate(eng.test_pairs().state())]
        eng.close()
        return out

    for ff in (True, False):
        a, b = run(False, ff), run(True, ff)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_native_iteration_with_side_information_equals_step_by_step(B, monkeypatch):
    """entity side information inside bdf_gibbs_sweep (uhat, per-row prior means, feature terms, beta: macau.jl:96-140 in one
    native call) gives bit for bit the chain of the iteration enqueued step by step from the host"""
    rng = np.random.default_rng(3)
    N1, N2, nnz, numF, D = 300, 120, 6000, 40, 16
    ids = np.stack([rng.integers(1, N1 + 1, nnz), rng.integers(1, N2 + 1, nnz)], axis=1)
    vals = np.clip(np.round(3.5 + rng.standard_normal(nnz)), 1, 5)
    F = rng.standard_normal((N1, numF))

    def run(ff):
        rel = B.Relation({"u": ids[:, 0], "v": ids[:, 1], "y": vals}, "r", [B.Entity("u", F=F), B.Entity("v")], dims=[N1, N2])
        B.setPrecision(rel, 1.5)
        rd = B.RelationData(rel)
        B.macau(rd, burnin=3, psamples=2, num_latent=D, verbose=False, seed=4, compute_ff_size=6500 if ff else 0)
        en = rd.entities[0]
        return rd._engine.native, en.model.sample.copy(), en.model.beta.copy(), float(en.lambda_beta), rd.entities[1].model.sample.copy()

    for ff in (True, False):
        n1 = run(ff)
        monkeypatch.setenv("BDF_NO_NATIVE", "1")
        n0 = run(ff)
        monkeypatch.delenv("BDF_NO_NATIVE")
        assert n1[0] is True and n0[0] is False
        assert np.array_equal(n1[1], n0[1]) and np.array_equal(n1[2], n0[2]) and n1[3] == n0[3] and np.array_equal(n1[4], n0[4])


def test_native_relation_model_equals_step_by_step(B, monkeypatch):
    """the relation model inside bdf_gibbs_sweep (macau.jl:83-92: sample_alpha, sample_beta_rel, linear_values before the rows;
    the test pairs' baseline mean + F_test beta refreshed for the prediction update) gives bit for bit the chain of the
    iteration enqueued step by step from the host -- alpha alone, relation features alone, both"""
    import pandas as pd
    rng = np.random.default_rng(3)
    A, Bm = rng.standard_normal((30, 2)), rng.standard_normal((40, 2))
    ii, jj = np.meshgrid(np.arange(1, 31), np.arange(1, 41), indexing="ij")
    feat = rng.standard_normal((1200, 2))
    v = (A @ Bm.T).ravel() + feat @ np.array([1.0, -1.0]) + 0.1 * rng.standard_normal(1200)

    def run(alpha_sample, with_feat):
        rd = B.RelationData(pd.DataFrame({"A": ii.ravel(), "B": jj.ravel(), "v": v}))
        r = rd.relations[0]
        r.model.alpha_sample = alpha_sample
        if with_feat:
            r.F = feat
        B.assignToTest(r, 10, rng=np.random.default_rng(1))
        res = B.macau(rd, burnin=6, psamples=5, num_latent=3, verbose=False, seed=9)
        return (rd._engine.native, rd.entities[0].model.sample.copy(), rd.entities[1].model.sample.copy(), float(r.model.alpha),
                None if not with_feat else np.asarray(r.model.beta).copy(), res["RMSE"], np.asarray(res["predictions"]["pred"]).copy())

    for alpha_sample, with_feat in ((True, False), (False, True), (True, True)):
        n1 = run(alpha_sample, with_feat)
        monkeypatch.setenv("BDF_NO_NATIVE", "1")
        n0 = run(alpha_sample, with_feat)
        monkeypatch.delenv("BDF_NO_NATIVE")
        assert n1[0] is True and n0[0] is False
        assert np.array_equal(n1[1], n0[1]) and np.array_equal(n1[2], n0[2]), (alpha_sample, with_feat)
        assert n1[3] == n0[3] and (not alpha_sample or n1[3] != 5.0)
        if with_feat:
            assert np.array_equal(n1[4], n0[4])
        assert n1[5] == n0[5] and np.array_equal(n1[6], n0[6])


def test_argument_errors(B):
    Y = _sprand(15, 10, 0.3, 1)
    rd = B.RelationData(Y)
    with pytest.raises(B.ArgumentError):
        B.macau(rd, burnin=1, psamples=1, verbose=False, output_beta=True)
    with pytest.raises(B.ArgumentError):
        B.macau(rd, burnin=1, psamples=1, verbose=False, output_type="hdf5")
    with pytest.raises(B.ArgumentError):
        B.macau(rd, burnin=1, psamples=1, verbose=False, num_latent=65)


def test_movielens_bpmf_quality(B):
    """BASELINE config 1/2 quality expectation: BPMF on the bundled MovieLens file, 500,000 held out, alpha 1.5,
    clamp [1,5]: RMSE ~ 0.862 at D=10 after 20+20; accept +-0.01."""
    from bdf_amd import datasets
    rd, source = datasets.movielens_relation_data(B)
    if source != "movielens_1m.mat":
        pytest.skip("bundled data file missing")
    res = B.macau(rd, burnin=20, psamples=20, num_latent=10, verbose=False, clamp=[1.0, 5.0], seed=3)
    assert abs(res["RMSE"] - 0.862) < 0.01, res["RMSE"]
    assert 0.84 < res["accuracy"] < 0.90
    assert 0.80 < res["ROC"] < 0.90


def test_movielens_macau_published_figure(B):
    """docs/index.md:42-88: Macau with the user and movie features, D=10, 100 burn-in + 400 samples, alpha 1.5:
    RMSE 0.8526, accuracy 0.8704, AUC 0.8485 (different split and random stream: +-0.004)."""
    from bdf_amd import datasets
    rd, source = datasets.movielens_relation_data(B, with_features=True)
    if source != "movielens_1m.mat":
        pytest.skip("bundled data file missing")
    res = B.macau(rd, burnin=100, psamples=400, num_latent=10, verbose=False, clamp=[1.0, 5.0], seed=11)
    assert abs(res["RMSE"] - 0.8526) < 0.004, res["RMSE"]
    assert abs(res["accuracy"] - 0.8704) < 0.004, res["accuracy"]
    assert abs(res["ROC"] - 0.8485) < 0.006, res["ROC"]
    assert res["lambda_beta"] > 0


def test_alpha_sampling(B):
    """test/alpha_sampling.jl:6-11"""
    Y = _sprand(15, 10, 0.3, 5)
    rd = B.RelationData(Y, class_cut=0.5, alpha_sample=True)
    B.assignToTest(rd.relations[0], 2, rng=np.random.default_rng(0))
    B.macau(rd, burnin=5, psamples=6, verbose=False)
    assert rd._engine.native               # sample_alpha runs inside the native iteration
    assert rd.relations[0].model.alpha > 0 and rd.relations[0].model.alpha != 1.0


def test_relation_features(B):
    """test/rel_feat.jl:6-28: fully observed 30 x 40 rank-2 relation plus two observation-level features, alpha sampled"""
    import pandas as pd
    rng = np.random.default_rng(3)
    A, Bm = rng.standard_normal((30, 2)), rng.standard_normal((40, 2))
    X = A @ Bm.T
    ii, jj = np.meshgrid(np.arange(1, 31), np.arange(1, 41), indexing="ij")
    df = pd.DataFrame({"A": ii.ravel(), "B": jj.ravel(), "v": X.ravel()})
    feat = rng.standard_normal((len(df), 2))
    beta = np.array([1.0, -1.0])
    df["v"] = df["v"] + feat @ beta
    rd = B.RelationData(df)
    rd.relations[0].model.alpha_sample = True
    rd.relations[0].F = feat
    B.assignToTest(rd.relations[0], 10, rng=np.random.default_rng(1))
    assert rd.relations[0].test_F.shape == (10, 2)
    result = B.macau(rd, burnin=50, psamples=10, num_latent=2, verbose=False)
    assert rd._engine.native               # sample_alpha, sample_beta_rel and linear_values run inside the native iteration
    # the model is exact (rank 2 + linear features): the relation beta is recovered and the held-out cells are predicted
    np.testing.assert_allclose(rd.relations[0].model.beta, beta, atol=0.1)
    assert result["RMSE"] < 0.3
    assert rd.relations[0].model.alpha > 1.0
    with pytest.raises(B.ArgumentError):
        B.macau(rd, burnin=1, psamples=1, num_latent=2, verbose=False, full_prediction=True)


def test_full_prediction(B):
    """macau.jl:37-39, 145-147, 228-230: predictions_full is the posterior mean of pred_all; on the test cells it equals
    the unclamped running mean of the test predictions"""
    Y = _sprand(20, 12, 0.4, 7)
    rd = B.RelationData(Y, class_cut=0.5)
    B.assignToTest(rd.relations[0], 15, rng=np.random.default_rng(2))
    res = B.macau(rd, burnin=3, psamples=5, num_latent=4, verbose=False, full_prediction=True, clamp=[])
    full = res["predictions_full"]
    assert full.shape == (20, 12)
    p = res["predictions"]
    ids = rd.relations[0].test_vec.ids
    np.testing.assert_allclose(full[ids[:, 0] - 1, ids[:, 1] - 1], np.asarray(p["pred"]), rtol=1e-10, atol=1e-12)


def test_movielens_d32_full_size_properties(B):
    """BASELINE config 2 at its full size (MovieLens, 500,209 training ratings, D=32): properties that do not need the CPU
    oracle at this size -- (a) two shards sample exactly the rows of the unsharded launch, bit for bit; (b) another item
    size gives the same rows to rounding; (c) sampled rows equal chol(inv(P_i))' z + inv(P_i) b_i recomputed in numpy from the
    row-system hook and the row's normals, on a sample of rows"""
    import ctypes as C
    from bdf_amd import datasets
    from bdf_amd._lib import check, lib
    rd, source = datasets.movielens_relation_data(B)
    if source != "movielens_1m.mat":
        pytest.skip("bundled data file missing")
    D = 32
    eng = B.GibbsEngine(rd, D, seed=7, device=0)
    for i in range(1, 4):
        eng.sweep(i)                    # a non-trivial state
    eng.sync()
    st = eng.ent[0]
    terms = eng._terms(0)
    p = lambda t: C.c_void_p(t.data_ptr())
    ctx = eng.ctx
    ctx.set_sweep(9)
    full = ctx.zeros(st.N, D)
    check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, p(st.mu), 0, p(st.Lambda), st.tag, 0, 1, p(full), None))
    halves = ctx.zeros(st.N, D)
    for s in (0, 1):
        check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, p(st.mu), 0, p(st.Lambda), st.tag, s, 2, p(halves), None))
    ctx.sync()
    a, b = full.cpu().numpy(), halves.cpu().numpy()
    assert np.array_equal(a, b)                                                       # (a)
    assert np.all(np.isfinite(a)) and 0.05 < np.abs(a).mean() < 5.0
    ctx.set_item_size(64)
    other = ctx.zeros(st.N, D)
    check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, p(st.mu), 0, p(st.Lambda), st.tag, 0, 1, p(other), None))
    ctx.sync()
    ctx.set_item_size(192)
    np.testing.assert_allclose(other.cpu().numpy(), a, rtol=1e-8, atol=1e-10)         # (b)
    # (c) the reference's map on a sample of rows
    P_t, b_t, z_t = ctx.zeros(st.N, D, D), ctx.zeros(st.N, D), ctx.zeros(st.N, D)
    check(lib().bdf_row_system(ctx.handle, D, st.N, 1, terms, p(st.mu), 0, p(st.Lambda), p(P_t), p(b_t)))
    check(lib().bdf_normals(ctx.handle, 1, st.tag, 0, st.N, D, p(z_t)))
    ctx.sync()
    P, bb, z = P_t.cpu().numpy(), b_t.cpu().numpy(), z_t.cpu().numpy()
    for row in (0, 1, 17, 1000, 3333, 6039):
        cov = np.linalg.inv(P[row])
        np.testing.assert_allclose(a[row], np.linalg.cholesky(cov) @ z[row] + cov @ bb[row], rtol=1e-8, atol=1e-9)
    assert ctx.rows_unfinished() == 0                     # every split row of every launch so far was finished
    eng.close()


def test_two_ranks_match_one(B, tmp_path):
    """bench.py's N > 1 path (rows shared out by bdf_layout_build, every rank holding its rows' observations only, in-place
    exchange of the sampled rows inside bdf_gibbs_sweep, test ratings split over the ranks, RMSE all-reduced) gives the chain
    of the single-process run of the same workload.  Two ranks on the one GPU of the box: RCCL needs a GPU per rank, so the
    small exchanges go through the library's host transport with a gloo all-gather behind it (BDF_DIST_BACKEND=gloo) -- and the
    large ones (the C4-shaped block's 5 MB per rank) by DIRECT PEER COPIES over IPC mappings (bdf_comm_enable_peer: two processes
    on one GPU can open each other's allocations), the transport the 8-GPU design needs for C4's 5 GB user factor."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BDF_DIST_BACKEND"] = "gloo"
    env["BDF_COMM_PEER"] = "1"          # (opt-in: unmeasured on a node with several GPUs)
    c5_sizes = "3000,16,60,40,120000,30000,400,8"
    # `python bench.py --gpus 2` as typed: bench.py starts its two ranks itself (torch's launcher, as a child process)
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "6", "--dump-state", str(tmp_path / "two"),
                          "--c4-rows", "20000", "--c4-cols", "3000", "--c4-nnz", "300000", "--c4-latent", "64", "--c5-sizes", c5_sizes, "--c5-sweeps", "6"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--replicas", "2", "--steps", "4", "--warmup", "6", "--no-cpu-baseline", "--no-c3", "--no-mref",
                          "--dump-state", str(tmp_path / "one"),
                          "--c4-rows", "20000", "--c4-cols", "3000", "--c4-nnz", "300000", "--c4-latent", "64", "--c5-sizes", c5_sizes, "--c5-sweeps", "6"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    d2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d2["config"]["units_per_sweep"] == 2 and d2["scaling"] == "weak"
    assert abs(d2["test_rmse"] - d1["test_rmse"]) < 2e-5, (d2["test_rmse"], d1["test_rmse"])
    # the strong-scaling block: the same C4-shaped relation on two ranks (rows at internal positions, sharded observations,
    # in-place exchange) and on one: the chains agree up to the summation order of the hyperprior's sums
    assert "error" not in d2["c4"] and "error" not in d1["c4"], (d2["c4"], d1["c4"])
    assert d2["c4"]["n_gpus"] == 2 and d2["c4"]["scaling"] == "strong"
    assert abs(d2["c4"]["test_rmse"] - d1["c4"]["test_rmse"]) < 1e-4, (d2["c4"], d1["c4"])
    ex = d2["c4"]["exchange"]
    assert "peer copies" in ex["transport"] and ex["peer_exchanges"] >= 10 and ex["peer_bytes_pulled_per_rank"] >= 10 * 5_000_000, ex
    # and the C5-shaped block (a shared entity with two relations and binary sparse features, beta's columns split over the ranks)
    assert "error" not in d2["c5"] and "error" not in d1["c5"], (d2["c5"], d1["c5"])
    assert d2["c5"]["n_gpus"] == 2 and d2["c5"]["beta_columns_per_rank"] == 16 and d2["c5"]["native_iteration"] and d1["c5"]["native_iteration"]
    assert abs(d2["c5"]["test_rmse"] - d1["c5"]["test_rmse"]) < 1e-5, (d2["c5"], d1["c5"])
    # ... and the chains themselves, element by element: every entity's sampled factor (the reference's row order), mu, Lambda
    # (and beta) of the three blocks -- the two ranks' replicas bit for bit after the exchange, the two-rank chain against the
    # single process's to 1e-9 (the hyperprior's sums are added in another order: rounding, carried through ten iterations)
    _compare_dumped_states(tmp_path, ("main", "c4", "c5"))
    # ... and the transport changes nothing: the same two ranks with the blocks staged through the host (BDF_COMM_PEER unset)
    # leave the C4-shaped block's factors bit for bit
    env.pop("BDF_COMM_PEER")
    host = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--dump-state", str(tmp_path / "host"),
                           "--no-c5", "--c4-rows", "20000", "--c4-cols", "3000", "--c4-nnz", "300000", "--c4-latent", "64"],
                          env=env, capture_output=True, text=True, timeout=600)
    assert host.returncode == 0, host.stdout[-2000:] + host.stderr[-2000:]
    dh = json.loads([l for l in host.stdout.splitlines() if l.startswith("{")][-1])
    assert "peer copies" not in dh["c4"]["exchange"]["transport"] and dh["c4"]["exchange"]["peer_exchanges"] == 0, dh["c4"]["exchange"]
    a, b = np.load(tmp_path / "two.c4.rank0.npz"), np.load(tmp_path / "host.c4.rank1.npz")
    for k in a.files:
        assert np.array_equal(a[k], b[k]), ("c4", k)


def _compare_dumped_states(tmp_path, blocks, rtol=1e-9):
    for block in blocks:
        r0, r1 = np.load(tmp_path / f"two.{block}.rank0.npz"), np.load(tmp_path / f"two.{block}.rank1.npz")
        o = np.load(tmp_path / f"one.{block}.rank0.npz")
        assert sorted(r0.files) == sorted(o.files) and len(r0.files) >= 6, (block, r0.files, o.files)
        for k in r0.files:
            assert np.array_equal(r0[k], r1[k]), (block, k)                  # one replica on every rank
            assert np.isfinite(r0[k]).all() and r0[k].shape == o[k].shape, (block, k)
            scale = max(1.0, float(np.abs(o[k]).max()))
            assert float(np.abs(r0[k] - o[k]).max()) <= rtol * scale, (block, k, float(np.abs(r0[k] - o[k]).max()), scale)


def test_two_ranks_match_one_with_four_rows_per_wave():
    """D = 10 / 12 with the short rows four to a wave (k_rows_small, forced for these small entities): rows at internal
    positions, a rank's own rows only, chunks -- the two-rank chains (host transport) equal the single process's, and both
    equal the wave-per-row kernel's chain to the reported digits"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--steps", "4", "--warmup", "6", "--num-latent", "10", "--no-cpu-baseline", "--no-c3", "--no-mref", "--no-c5",
            "--c4-rows", "20000", "--c4-cols", "3000", "--c4-nnz", "300000", "--c4-latent", "12"]
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = {}
    for name, extra, env in (("two", ["--gpus", "2"], dict(base, BDF_DIST_BACKEND="gloo", BDF_K1_SMALL_MIN_ROWS="1")),
                             ("one", ["--replicas", "2"], dict(base, BDF_K1_SMALL_MIN_ROWS="1")),
                             ("old", ["--replicas", "2"], dict(base, BDF_K1_SMALL="0"))):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra + args, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        out[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert "error" not in out[name]["c4"], out[name]["c4"]
    for name in ("one", "old"):
        assert abs(out["two"]["test_rmse"] - out[name]["test_rmse"]) < 2e-5, (name, out["two"]["test_rmse"], out[name]["test_rmse"])
        assert abs(out["two"]["c4"]["test_rmse"] - out[name]["c4"]["test_rmse"]) < 1e-4, (name, out["two"]["c4"], out[name]["c4"])


@pytest.mark.parametrize("mode", ["plain", "rccl"])
def test_stream_schedule_soak(B, mode):
    """two runs of 1,500 sweeps of the bench workload (three streams, rows rotating through three buffers, the row kernels
    polling for the hyperprior draws, prediction updates beside the rows) end bit-identical and leave no split row unfinished --
    tools/soak_determinism.py runs 20,000.  rccl: the same with a ONE-rank RCCL communicator in the iteration -- ncclAllGather
    kernels on the device between the row launches AND inside the hyperprior's sums (bdf_hyper_sums_ranks through the
    collective), the row kernels still polling (BDF_POLL_WITH_COMM) -- twice, bit-identical, and equal to rounding to the run
    without a communicator; no spin bound hit (flag 16 would raise at the next synchronisation)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BDF_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_determinism.py"), "1500"] + (["rccl"] if mode == "rccl" else []),
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bit-identical: True" in r.stdout and "unfinished=0" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    if mode == "rccl":
        assert r.stdout.count("communicator=RCCL") == 2 and "communicator=None" in r.stdout, r.stdout[-1500:]
