"""A larger synthetic relation on one GPU: set-up times, sweep time, determinism under a different item size.
   python3 tools/scale_probe.py [n_users n_items nnz D]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bdf_amd as B

nu, ni, nnz, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1_000_000, 100_000, 20_000_000, 32)))
rng = np.random.default_rng(777)
t0 = time.time()
rows = rng.integers(1, nu + 1, nnz)
# item popularity ~ 1 / (rank + 100)
p = 1.0 / (np.arange(ni) + 100.0); p /= p.sum()
cols = rng.choice(ni, size=nnz, p=p) + 1
# planted rank-8 model
Us, Vs = rng.standard_normal((nu, 8)) * 0.5, rng.standard_normal((ni, 8)) * 0.5
vals = np.clip(np.round(3.5 + np.einsum("ij,ij->i", Us[rows - 1], Vs[cols - 1]) + 0.5 * rng.standard_normal(nnz)), 1, 5)
print(f"generated {nnz} observations in {time.time() - t0:.1f}s")
t0 = time.time()
idf = B.IndexedDF((np.stack([rows, cols], axis=1), vals), [nu, ni])
rd = B.RelationData(idf, class_cut=2.5, alpha=2.0)
print(f"IndexedDF + RelationData in {time.time() - t0:.1f}s")
t0 = time.time()
eng = B.GibbsEngine(rd, D, seed=5, device=0)
eng.sweep(1); eng.sync()
print(f"engine set-up + first sweep (plans) in {time.time() - t0:.1f}s; "
      f"device memory {torch.cuda.memory_allocated() / 2**30:.2f} GiB (torch) ")
for i in range(2, 5):
    eng.sweep(i)
eng.sync()
t0 = time.time()
n = 10
for i in range(5, 5 + n):
    eng.sweep(i)
eng.sync()
dt = (time.time() - t0) / n
bytes_sweep = sum(eng.k1_algorithmic_bytes(j) for j in range(2))
print(f"sweep {dt * 1e3:.2f} ms  ({1 / dt:.1f} sweeps/s), algorithmic {bytes_sweep / 1e9:.2f} GB/sweep -> {bytes_sweep / dt / 1e12:.2f} TB/s")
s1 = [st.sample.clone() for st in eng.ent]
finite = all(bool(torch.isfinite(s).all()) for s in s1)
print("finite:", finite, " row norms:", [float(s.norm(dim=1).mean()) for s in s1])
eng.close()
