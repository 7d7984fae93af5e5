"""Does the row kernel's speed depend on how far the chain has run (GPU box)?  Fresh engine; after n iterations in total:
the two row launches alone, back to back, 200 pairs (bdf_gibbs_rows_only: inputs are the chain's current samples and prior),
and the pace of 20 full iterations."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import check, lib

rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1)
test = eng.test_pairs()
eng.warm_device(60.0)
it = 0
k = 0
def alone():
    global k
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        k += 1
        check(lib().bdf_gibbs_rows_only(eng.gibbs, 0, 1_000_000 + 2 * k))
        check(lib().bdf_gibbs_rows_only(eng.gibbs, 1, 1_000_001 + 2 * k))
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / 200
def region(n):
    global it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        it += 1
        eng.step(it, 2 if it > 1 else 1, [1.0, 5.0], rel.class_cut)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
for target in (5, 25, 50, 100, 200, 400, 800):
    pace = region(target - it) if target > it else float("nan")
    a = alone()
    r = region(20)
    import numpy as np
    U = eng.ent[0].sample
    print(f"after {it - 20:4d} iterations: pair of row launches alone {a:6.1f} us; the next 20 iterations {r:6.1f} us each; "
          f"|U| mean abs {float(U.abs().mean()):.3f}, Lambda_U diag mean {float(torch.diagonal(eng.ent[0].Lambda).mean()):.2f}")
eng.close()
