"""Main-stream timeline of a few sweeps from HIP events (no profiler attached): where the stream idles between its kernels."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
test = eng.test_pairs()
main = eng.ctx.stream
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(main); marks.append((name, e))
orig_sample = eng.sample_entity
def sample_entity(j):
    mark(f"rows{j}<"); orig_sample(j); mark(f"rows{j}>")
eng.sample_entity = sample_entity
def step(i, phase, m=False):
    eng.sweep(i)
    if m: mark("pred<")
    test.update(32, eng.factors_of(rel), rel.model.mean_value, phase, [1.0, 5.0], rel.class_cut)
    if m: mark("pred>")
for i in range(1, 31):
    step(i, 0)
eng.sync(); torch.cuda.synchronize(); marks.clear()
for k in range(6):
    step(100 + k, 2, True)
eng.sync(); torch.cuda.synchronize()
t0 = marks[0][1]
prev = 0.0
for name, e in marks[6 * 2:]:
    t = t0.elapsed_time(e) * 1e3
    print(f"{name:8s} {t:9.1f} us  (+{t - prev:6.1f})")
    prev = t
