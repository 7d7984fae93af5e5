"""What the timed region of bench.py costs beyond the sweeps (GPU box): 200 steps with and without K1 events every 8th sweep,
the host's enqueue time next to the total, the cost of creating a timer pair."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd.engine import KernelTimer
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
test = eng.test_pairs()
for i in range(1, 501):
    eng.step(i, 0, [1.0, 5.0], rel.class_cut)
eng.sync()
t0 = time.perf_counter()
timers = [KernelTimer() for _ in range(100)]
print(f"KernelTimer(): {1e6 * (time.perf_counter() - t0) / 100:.1f} us each")
it = 1000
for rep in range(3):
    for every in (0, 8):
        eng.k1_events = [] if every else None
        eng.k1_event_every = every or 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(200):
            it += 1
            eng.step(it, 2, [1.0, 5.0], rel.class_cut)
        t1 = time.perf_counter()
        eng.sync(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"200 steps, K1 events every {every}: enqueue {1e6 * (t1 - t0) / 200:.1f} us/step, total {1e6 * (t2 - t0) / 200:.1f} us/step")
        eng.k1_events = None
eng.close()
