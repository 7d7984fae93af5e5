"""Host-side cost of enqueuing one sweep (+ prediction update): time the enqueue loop alone, then the whole thing."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
test = eng.test_pairs()
def step(i, phase):
    eng.sweep(i)
    test.update(32, eng.factors_of(rel), rel.model.mean_value, phase, [1.0, 5.0], rel.class_cut)
for i in range(1, 21):
    step(i, 0)
eng.sync(); torch.cuda.synchronize()
for n in (20, 100):
    t0 = time.perf_counter()
    for k in range(n):
        step(100 + k, 2)
    t1 = time.perf_counter()
    eng.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: enqueue {1e6 * (t1 - t0) / n:.1f} us/step, total {1e6 * (t2 - t0) / n:.1f} us/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(200):
    step(300 + k, 2)
pr.disable(); eng.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
