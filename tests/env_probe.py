"""A fresh process for the parity tests of library switches that are read once per process (environment variables):
the caller sets the environment, this script runs one small scenario through the C ABI and leaves its arrays in an .npz file;
the test compares them with the CPU oracle.  usage: env_probe.py <scenario> <out.npz> [args...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B


def macau_problem(seed, N1, N2, nnz):
    rng = np.random.default_rng(seed)
    ids = np.stack([rng.integers(1, N1 + 1, nnz), rng.integers(1, N2 + 1, nnz)], axis=1)
    vals = rng.standard_normal(nnz)
    return ids, vals


def lowrank_problem(seed, dims, D):
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 17, dims[0])
    deg[:9] = [0, 1, 2, 7, 14, 15, 3, 40, 16]
    rows = np.repeat(np.arange(1, dims[0] + 1), deg)
    ids = np.stack([rows, rng.integers(1, dims[1] + 1, len(rows))], axis=1).astype(np.int64)
    vals = rng.random(len(rows)) * 4 + 1
    facs = [rng.standard_normal((d, D)) * 0.5 for d in dims]
    A = rng.standard_normal((D, D))
    return ids, vals, facs, A @ A.T / D + np.eye(D), rng.standard_normal(D), rng.standard_normal((dims[0], D))


def main():
    scenario, out = sys.argv[1], sys.argv[2]
    if scenario == "macau":
        seed, N1, N2, nnz, D, iters = (int(x) for x in sys.argv[3:9])
        ids, vals = macau_problem(seed, N1, N2, nnz)
        rel = B.Relation({"u": ids[:, 0], "v": ids[:, 1], "y": vals}, "r", [B.Entity("u"), B.Entity("v")], dims=[N1, N2])
        B.setPrecision(rel, 2.0)
        rd = B.RelationData(rel)
        B.macau(rd, burnin=iters, psamples=0, num_latent=D, verbose=False, seed=77)
        np.savez(out, **{f"S{j}": rd.entities[j].model.sample.T for j in (0, 1)}, **{f"mu{j}": rd.entities[j].model.mu for j in (0, 1)},
                 **{f"Lam{j}": rd.entities[j].model.Lambda for j in (0, 1)})
    elif scenario == "lowrank":
        import ctypes as C
        from bdf_amd._lib import Term, check, lib
        seed, n_rows, D = (int(x) for x in sys.argv[3:6])
        dims = [n_rows, 60]
        ids, vals, facs, Lam, mu, mu_rows = lowrank_problem(seed, dims, D)
        ctx = B.Context(seed=1234)
        dr = B.DeviceRelation(ctx, B.IndexedDF((ids, vals), dims))
        ft = ctx.tensor(facs[1])
        terms = (Term * 1)()
        t = terms[0]
        t.rel = dr.handle; t.mode = 0; t.alpha = 1.7; t.mean_value = float(vals.mean()); t.linear_values = None
        t.factors[0] = None; t.factors[1] = ft.data_ptr()
        ctx.set_lowrank(-1, 0)
        ctx.set_sweep(7)
        res = {}
        for name, m in (("shared", mu), ("per_row", mu_rows)):
            out_t, mu_t, Lam_t = ctx.zeros(dims[0], D), ctx.tensor(m), ctx.tensor(Lam)
            guard = ctx.zeros(8 * D)                                # (the padding records of the old bug wrote before the matrix)
            check(lib().bdf_sample_rows(ctx.handle, D, dims[0], 1, terms, C.c_void_p(mu_t.data_ptr()), int(m.ndim == 2),
                                        C.c_void_p(Lam_t.data_ptr()), 5, 0, 1, C.c_void_p(out_t.data_ptr()), None))
            ctx.sync()
            res[name] = out_t.cpu().numpy()
            res[name + "_guard"] = guard.cpu().numpy()
        d = ctx.rows_dispatch(5)
        np.savez(out, lr_rows=np.array([d["lowrank"]]), **res)
        dr.close(); ctx.close()
    else:
        raise SystemExit(f"unknown scenario {scenario}")


if __name__ == "__main__":
    main()
