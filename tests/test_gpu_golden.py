"""The device path against the committed golden vectors (tests/golden/*.npz): row systems and conditional means from an
independent numpy computation, the index literal of the reference's test/basic.jl, the Philox stream contract."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def test_device_rows_against_golden(B, ctx):
    from bdf_amd._lib import Term, check, lib
    g = np.load(os.path.join(G, "rows_small.npz"))
    D, N = int(g["D"]), int(g["dimsA"][0])
    drA = B.DeviceRelation(ctx, B.IndexedDF((g["idsA"], g["vA"]), list(g["dimsA"])))
    drB = B.DeviceRelation(ctx, B.IndexedDF((g["idsB"], g["vB"]), list(g["dimsB"])))
    FA1, FA2, FB1 = ctx.tensor(g["FA1"]), ctx.tensor(g["FA2"]), ctx.tensor(g["FB1"])
    terms = (Term * 2)()
    terms[0].rel = drA.handle; terms[0].mode = 0; terms[0].alpha = float(g["alphaA"]); terms[0].mean_value = float(g["meanA"])
    terms[0].factors[1] = FA1.data_ptr(); terms[0].factors[2] = FA2.data_ptr()
    terms[1].rel = drB.handle; terms[1].mode = 0; terms[1].alpha = float(g["alphaB"]); terms[1].mean_value = float(g["meanB"])
    terms[1].factors[1] = FB1.data_ptr()
    mu_t, Lam_t = ctx.tensor(g["mu"]), ctx.tensor(g["Lambda"])
    P_t, b_t = ctx.zeros(N, D, D), ctx.zeros(N, D)
    check(lib().bdf_row_system(ctx.handle, D, N, 2, terms, _p(mu_t), 0, _p(Lam_t), _p(P_t), _p(b_t)))
    ctx.sync()
    np.testing.assert_allclose(P_t.cpu().numpy(), g["P"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b_t.cpu().numpy(), g["b"], rtol=1e-12, atol=1e-12)
    # the sampled rows with the device's own normals: mean part = golden conditional mean, noise part = chol(inv(P))' z
    out = ctx.zeros(N, D)
    ctx.set_sweep(3)
    check(lib().bdf_sample_rows(ctx.handle, D, N, 2, terms, _p(mu_t), 0, _p(Lam_t), 4, 0, 1, _p(out), None))
    z = ctx.zeros(N, D)
    check(lib().bdf_normals(ctx.handle, 1, 4, 0, N, D, _p(z)))
    ctx.sync()
    x, zz = out.cpu().numpy(), z.cpu().numpy()
    for row in range(N):
        cov = np.linalg.inv(g["P"][row])
        np.testing.assert_allclose(x[row], np.linalg.cholesky(cov) @ zz[row] + g["mean"][row], rtol=1e-9, atol=1e-10)
    drA.close(); drB.close()


def test_device_index_against_reference_literal(B, ctx):
    g = np.load(os.path.join(G, "index_basic.npz"))
    dr = B.DeviceRelation(ctx, B.IndexedDF((g["ids"], g["values"]), list(g["dims"])))
    for mode in (0, 1):
        rp, ri = dr.index(mode)
        assert np.array_equal(rp, g[f"rowptr{mode}"])
        assert np.array_equal(ri, g[f"rowids{mode}"])             # 1-based table rows, in table order (IndexedDF.jl:10-21)
    dr.close()


def test_device_stream_against_golden(B):
    from bdf_amd._lib import check, lib
    g = np.load(os.path.join(G, "philox.npz"))
    out = (C.c_uint32 * 4)()
    for c, (s, sw, p, e, r) in enumerate(g["cases"]):
        cx = B.Context(seed=int(s))
        cx.set_sweep(int(sw))
        for pair in range(4):
            check(lib().bdf_philox(cx.handle, int(p), int(e), int(r), pair, out))
            assert list(out) == list(g["draws"][c, pair])
        z = cx.zeros(1, 8)
        check(lib().bdf_normals(cx.handle, int(p), int(e), int(r), 1, 8, _p(z)))
        cx.sync()
        np.testing.assert_allclose(z.cpu().numpy()[0], g["normals"][c], rtol=1e-13, atol=1e-14)
        cx.close()
