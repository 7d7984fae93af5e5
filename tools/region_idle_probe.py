"""How much does the idle time in front of a 20-iteration region cost (GPU box)?  Engine warm-up (60 ms), then repeatedly:
200 untimed iterations (sustained load), full synchronisation, an idle pause of X, 20 timed iterations, synchronisation."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets

rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1)
test = eng.test_pairs()
eng.warm_device(60.0)
it = 0
def steps(n, phase):
    global it
    for k in range(n):
        it += 1
        eng.step(it, phase, [1.0, 5.0], rel.class_cut)
for pre in (200, 5):
    for pause_us in (0, 50, 200, 1000, 5000, 20000, 0):
        res = []
        for rep in range(4):
            steps(pre, 0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            while 1e6 * (time.perf_counter() - t1) < pause_us:
                pass
            t0 = time.perf_counter()
            steps(20, 2)
            torch.cuda.synchronize()
            res.append(1e6 * (time.perf_counter() - t0) / 20)
        print(f"{pre:4d} iterations, sync, pause {pause_us:6d} us, then 20 timed iterations: " + " ".join(f"{x:.1f}" for x in res) + " us per iteration")
eng.close()
