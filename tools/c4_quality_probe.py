"""C4-shaped relation (1M x 100k, 20M observations, D = 64): held-out RMSE after B + S sweeps (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd.engine import GibbsEngine
rd = datasets.c4_relation_data(B, 1_000_000, 100_000, 20_000_000)
rel = rd.relations[0]
tv = np.asarray(rel.test_vec.values)
for nb in (10, 30, 60):
    eng = GibbsEngine(rd, 64, seed=5)
    test = eng.test_pairs()
    t0 = time.time()
    for i in range(1, 2 * nb + 1):
        stats = eng.step(i, 0 if i <= nb else (1 if i == nb + 1 else 2), [1.0, 5.0], rel.class_cut)
    eng.sync()
    print(f"{nb}+{nb}: rmse {float(np.sqrt(stats.cpu().numpy()[0] / test.n)):.4f} (std of the held-out values {tv.std():.4f}) in {time.time() - t0:.1f} s", flush=True)
    eng.close()
