"""host time to enqueue a sweep (native bdf_gibbs_sweep) against the GPU time per sweep (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
for i in range(1, 201):
    eng.step(i, 0, [1.0, 5.0], rel.class_cut)
eng.sync()
for n in (50, 400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        eng.step(1000 + i, 2, [1.0, 5.0], rel.class_cut)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print(f"{n} sweeps: enqueue {1e6 * (t1 - t0) / n:.1f} us/sweep, total {1e6 * (t2 - t0) / n:.1f} us/sweep")
eng.close()
