"""The reference's own benchmark shape (test/benchmark_parallel_latent.jl:8-61): sprand(1_500_000, 1000, 0.01)-like relation
(~15M uniform positions, U(0,1) values, seed 1500), BPMF D=10 and D=30, 2+2 iterations; ms per sweep on one GPU and the
algorithmic GB/s of the row kernel's share (SURVEY 8d bytes)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd.engine import GibbsEngine

rng = np.random.default_rng(1500)
N, M = 1_500_000, 1000
nnz = int(N * M * 0.01)
key = np.unique(rng.integers(0, N * M, size=int(nnz * 1.01)))[:nnz]
ids = np.stack([key // M + 1, key % M + 1], axis=1)
vals = rng.random(len(key))
# MREF_D=10 or 30: one D per process (a profile of the run then names THAT configuration's dominant kernel)
for D in ([int(os.environ['MREF_D'])] if os.environ.get('MREF_D') else (10, 30)):
    t0 = time.time()
    rel = B.Relation((ids, vals), "r", [B.Entity("rows"), B.Entity("cols")], dims=[N, M])
    B.assignToTest(rel, np.arange(1, 51))
    rd = B.RelationData(rel)
    eng = GibbsEngine(rd, D, seed=1)
    eng.sync()
    setup = time.time() - t0
    for i in range(1, 3):
        eng.sweep(i)
    eng.sync()
    t0 = time.time()
    n = 10
    for i in range(3, 3 + n):
        eng.sweep(i)
    eng.sync()
    dt = (time.time() - t0) / n
    bytes_sweep = sum(eng.k1_algorithmic_bytes(j) for j in (0, 1))
    print(f"M-ref D={D}: {len(key)} observations, set-up {setup:.1f} s, {dt * 1e3:.2f} ms/sweep, "
          f"{bytes_sweep / dt / 1e9:.0f} GB/s algorithmic")
    eng.close()
