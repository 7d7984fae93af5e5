"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from an
independent numpy computation / the reference's test literals)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _terms(O, g):
    tA = O.Term(g["idsA"], g["vA"], list(g["dimsA"]), 0, float(g["alphaA"]), float(g["meanA"]), [None, g["FA1"], g["FA2"]])
    tB = O.Term(g["idsB"], g["vB"], list(g["dimsB"]), 0, float(g["alphaB"]), float(g["meanB"]), [None, g["FB1"]])
    return [tA, tB]


def test_oracle_rows_against_golden(O):
    g = np.load(os.path.join(G, "rows_small.npz"))
    D = int(g["D"])
    terms = _terms(O, g)
    for row in range(int(g["dimsA"][0])):
        P, b = O.row_system(D, terms, row, g["mu"], g["Lambda"])
        np.testing.assert_allclose(P, g["P"][row], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(b, g["b"][row], rtol=1e-12, atol=1e-12)
        x, m = O.sample_row(D, terms, row, g["mu"], g["Lambda"], g["z"][row])
        np.testing.assert_allclose(m, g["mean"][row], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(x, g["x"][row], rtol=1e-10, atol=1e-12)


def test_oracle_normal_wishart_parameters_against_golden(O):
    g = np.load(os.path.join(G, "nw_small.npz"))
    mu_N, beta_N, T_N, nu_N = O.hyper_params(g["U"], g["mu0"], float(g["b0"]), g["Tinv"], float(g["nu"]))
    np.testing.assert_allclose(mu_N, g["mu_N"], rtol=1e-12)
    assert beta_N == float(g["beta_N"]) and nu_N == float(g["nu_N"])
    np.testing.assert_allclose(T_N, g["T_N"], rtol=1e-9, atol=1e-12)


def test_oracle_index_against_reference_literal(O):
    g = np.load(os.path.join(G, "index_basic.npz"))
    idx = O.index_build(g["ids"], list(g["dims"]))
    for mode in (0, 1):
        rp, ri = idx[mode]
        assert np.array_equal(rp, g[f"rowptr{mode}"])
        assert np.array_equal(ri[:len(g[f"rowids{mode}"])], g[f"rowids{mode}"])


def test_oracle_stream_against_golden(O):
    g = np.load(os.path.join(G, "philox.npz"))
    for c, (s, sw, p, e, r) in enumerate(g["cases"]):
        for pair in range(4):
            assert list(O.draw(int(s), int(sw), int(p), int(e), int(r), pair)) == list(g["draws"][c, pair])
        assert np.array_equal(O.normals(int(s), int(sw), int(p), int(e), int(r), 8), g["normals"][c])
