"""in-kernel launch spans (bdf_gibbs_span_rows) against the wall clock: row launches alone, back to back, and inside sweeps (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import check, lib
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
eng.test_pairs(); eng.register_test([1.0, 5.0], 2.5)
for i in range(1, 60):
    eng.sweep(i, 2)
eng.sync()
import torch
n = 200
eng.k1_span_begin(n)
t0 = time.perf_counter()
for i in range(n // 2):
    eng.sweep(1000 + i, 2)
eng.sync()
wall = 1e6 * (time.perf_counter() - t0) / (n // 2)
tt, ents = eng.k1_spans
h = tt.cpu().numpy().view(np.uint64)
st, en = h[:, :, 0].min(axis=1).astype(np.int64), h[:, :, 1].max(axis=1).astype(np.int64)
d = (en - st) / 100.0
gap = (st[1:] - en[:-1]) / 100.0
print(f"in sweeps: wall {wall:.2f} us per sweep; span mean {d.mean():.2f} (users {d[0::2].mean():.2f}, movies {d[1::2].mean():.2f}) min {d.min():.2f} max {d.max():.2f}; "
      f"end of one -> start of the next: mean {gap.mean():.2f} min {gap.min():.2f} max {gap.max():.2f}")
eng.close()
