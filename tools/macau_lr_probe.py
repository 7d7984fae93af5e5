"""Macau at the shape side information is for -- many rows with a handful of observations and a feature vector each -- with and
without the low-rank sampler (BDF_LOWRANK=0 in a second process): ms per sweep (GPU box).
   python tools/macau_lr_probe.py [N1 N2 nnz D numF]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd.engine import GibbsEngine
N1, N2, nnz, D, numF = [int(x) for x in (sys.argv[1:6] if len(sys.argv) >= 6 else "100000 2000 1000000 32 64".split())]
rng = np.random.default_rng(24)
key = np.unique(rng.integers(0, N1 * N2, size=int(nnz * 1.02)))[:nnz]
ids = np.stack([key // N2 + 1, key % N2 + 1], axis=1)
F = rng.standard_normal((N1, numF))
W = rng.standard_normal((numF, 3)) * 0.4
vals = np.sum((F @ W)[ids[:, 0] - 1] * rng.standard_normal((N2, 3))[ids[:, 1] - 1], axis=1) + 0.3 * rng.standard_normal(nnz)
rel = B.Relation((ids, vals), "r", [B.Entity("compounds", F=F), B.Entity("proteins")], dims=[N1, N2])
B.setPrecision(rel, 2.0)
rd = B.RelationData(rel)
eng = GibbsEngine(rd, D, seed=31)
for i in range(1, 11):
    eng.sweep(i)
eng.sync()
t0 = time.perf_counter()
n = 40
for i in range(11, 11 + n):
    eng.sweep(i)
eng.sync()
print(f"Macau {N1} x {N2}, {nnz} observations, {numF} features, D={D}, BDF_LOWRANK={os.environ.get('BDF_LOWRANK', '(default)')}: "
      f"{1e3 * (time.perf_counter() - t0) / n:.3f} ms per sweep, low-rank rows {[eng.lowrank_rows(j) for j in (0, 1)]}", flush=True)
eng.close()
