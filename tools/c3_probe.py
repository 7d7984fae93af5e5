"""Config C3: Macau on MovieLens + dense user side information 6040 x 500, D=32; FF path and forced CG."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bdf_amd as B
from bdf_amd import datasets

F = np.random.default_rng(4242).standard_normal((6040, 500))
which = sys.argv[1] if len(sys.argv) > 1 else "both"          # ff | cg | both
for ff_size, label in ((6500, "FF path (numF=500 <= compute_ff_size)"), (0, "CG forced (compute_ff_size=0)")):
    if which != "both" and (which == "ff") != (ff_size > 0):
        continue
    rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
    rd.entities[0].F = F
    eng = B.GibbsEngine(rd, 32, seed=1, device=0, compute_ff_size=ff_size)
    for i in range(1, 6):
        eng.sweep(i)
    eng.sync()
    t0 = time.time()
    n = 20
    for i in range(6, 6 + n):
        eng.sweep(i)
    eng.sync()
    dt = (time.time() - t0) / n
    it = eng.ent[0].cg_iters.cpu().numpy()
    print(f"{label}: {dt * 1e3:.2f} ms/sweep ({1 / dt:.0f} sweeps/s); CG iterations per column (last sweep): min {it.min()} max {it.max()}; "
          f"lambda_beta {float(eng.ent[0].lambda_beta.item()):.2f}")
    eng.close()
