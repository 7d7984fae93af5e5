"""Sweep time of the bench workload for every (side, prediction) pair out of the first six streams of torch's pool: how much
the mapping of HIP streams onto hardware queues matters (the engine keeps whatever pair its first engine got)."""
import os, sys, time, itertools
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets, engine as E
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
dev = torch.device("cuda", 0)
S = [torch.cuda.Stream(dev) for _ in range(6)]
def run(a, b, n=300):
    E._SIDE_STREAMS[(0, 0)], E._SIDE_STREAMS[(0, 1)] = S[a], S[b]
    eng = B.GibbsEngine(rd, 32, seed=1, device=0)
    got = (eng.ctx_h.stream is S[a], eng.ctx_p.stream is S[b])
    test = eng.test_pairs()
    for i in range(1, 201):
        eng.sweep(i); test.update(32, eng.factors_of(rel), rel.model.mean_value, 0, [1.0, 5.0], rel.class_cut)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(201, 201 + n):
        eng.sweep(i); test.update(32, eng.factors_of(rel), rel.model.mean_value, 2, [1.0, 5.0], rel.class_cut)
    eng.sync(); torch.cuda.synchronize()
    dt = 1e6 * (time.perf_counter() - t0) / n
    eng.close()
    return dt, got
run(0, 1)
for a, b in itertools.permutations(range(6), 2):
    dt, got = run(a, b)
    print(f"side=S{a} pred=S{b}: {dt:6.1f} us/sweep {'' if all(got) else '(self-test chose other streams)'}")
