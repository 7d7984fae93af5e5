"""M-ref at D = 30 only: 2 + 6 iterations, ms per sweep (run under rocprofv3 --kernel-trace for the per-launch durations)."""
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
D = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rel = B.Relation((ids, vals), "r", [B.Entity("rows"), B.Entity("cols")], dims=[N, M])
B.assignToTest(rel, np.arange(1, 51))
rd = B.RelationData(rel)
eng = GibbsEngine(rd, D, seed=1)
eng.sync()
for i in range(1, 3): eng.sweep(i)
eng.sync()
t0 = time.time()
n = 6
for i in range(3, 3 + n): eng.sweep(i)
eng.sync()
print(f"M-ref D={D} BDF_K1_COL={os.environ.get('BDF_K1_COL')}: {(time.time() - t0) / n * 1e3:.3f} ms/sweep", flush=True)
eng.close()
