"""Config C5 on ONE GPU: entity A (100,000) shared by a 3-mode relation A x B x C (B=64, C=1,000; 5M cells
from a planted rank-8 CP model, seed 901) and a 2-mode relation A x T (T=500; 1M cells, seed 902); A has binary sparse
features 100,000 x 50,000 with 50 nnz per row (seed 903, binary CSR); Macau D=32, alpha 5 / 2, CG on the binary features;
1% of the first relation held out.  Prints ms per sweep and the held-out RMSE against the generator's noise (0.1)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
import bdf_amd as B

nA, nB, nC, nT, R = 100_000, 64, 1_000, 500, 8
rng = np.random.default_rng(901)
fa, fb, fc = rng.standard_normal((nA, R)) * 0.7, rng.standard_normal((nB, R)) * 0.7, rng.standard_normal((nC, R)) * 0.7
n1 = 5_000_000
key = np.unique(rng.integers(0, nA * nB * nC, size=int(n1 * 1.02)))[:n1]
ia, ib, ic = key // (nB * nC), (key // nC) % nB, key % nC
v1 = np.sum(fa[ia] * fb[ib] * fc[ic], axis=1) + 0.1 * rng.standard_normal(len(key))
rng2 = np.random.default_rng(902)
ft = rng2.standard_normal((nT, R)) * 0.7
n2 = 1_000_000
key2 = np.unique(rng2.integers(0, nA * nT, size=int(n2 * 1.02)))[:n2]
ja, jt = key2 // nT, key2 % nT
v2 = np.sum(fa[ja] * ft[jt], axis=1) + 0.1 * rng2.standard_normal(len(key2))
rng3 = np.random.default_rng(903)
cols = rng3.integers(0, 50_000, size=(nA, 50))
Fbin = sp.csr_matrix((np.ones(nA * 50), (np.repeat(np.arange(nA), 50), cols.ravel())), shape=(nA, 50_000))
Fbin.data[:] = 1.0

A = B.Entity("A", F=Fbin)
Bn, Cn, Tn = B.Entity("B"), B.Entity("C"), B.Entity("T")
r1 = B.Relation((np.stack([ia + 1, ib + 1, ic + 1], axis=1), v1), "abc", [A, Bn, Cn], dims=[nA, nB, nC])
r2 = B.Relation((np.stack([ja + 1, jt + 1], axis=1), v2), "at", [A, Tn], dims=[nA, nT])
B.assignToTest(r1, n1 // 100, rng=np.random.default_rng(5))
B.setPrecision(r1, 5.0)
B.setPrecision(r2, 2.0)
rd = B.RelationData()
B.addRelation(rd, r1)
B.addRelation(rd, r2)
assert len(A.relations) == 2
t0 = time.time()
nb = int(os.environ.get("C5_BURNIN", "10"))
res = None if os.environ.get("C5_SWEEPS_ONLY") else B.macau(rd, burnin=nb, psamples=nb, num_latent=32, verbose=False, compute_ff_size=0, seed=3)
wall = time.time() - t0
if res is not None:
  print(f"C5 on one GPU: {2 * nb} sweeps + set-up in {wall:.1f} s; held-out RMSE of relation abc {res['RMSE']:.4f} (noise 0.1, "
      f"value std {v1.std():.3f}); lambda_beta {A.lambda_beta:.2f}")
from bdf_amd.engine import GibbsEngine
eng = GibbsEngine(rd, 32, seed=3, compute_ff_size=0)
for i in range(1, 4):
    eng.sweep(i)
eng.sync()
t0 = time.time()
for i in range(4, 14):
    eng.sweep(i)
eng.sync()
dt = (time.time() - t0) / 10
it = eng.ent[0].cg_iters.cpu().numpy()
print(f"C5 sweep (rows of A, B, C, T + hyperpriors + beta of A by CG on the binary features): {dt * 1e3:.2f} ms; "
      f"CG iterations per column min {it.min()} max {it.max()}")
eng.close()
