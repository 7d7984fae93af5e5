"""What crosses the host boundary (GPU box): the reference hands over host arrays once -- Relation.data (ids, values) -- and
reads samples back when it wants them; nothing crosses per iteration.  Times on the bench workload (MovieLens-1M, D = 32):
engine set-up (index build on the host + upload + plans), one iteration, and reading both factor matrices back to the host
(what a host that wants every sample on its side, PCIe included, would pay per iteration)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bdf_amd as B
from bdf_amd import datasets

rd, source = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter()
eng = B.GibbsEngine(rd, 32, seed=1)
test = eng.test_pairs()
eng.sync(); torch.cuda.synchronize()
t_setup = time.perf_counter() - t0
eng.warm_device(60.0)
for i in range(1, 201):
    eng.step(i, 0, [1.0, 5.0], rel.class_cut)
eng.sync()
t0 = time.perf_counter()
for i in range(201, 401):
    eng.step(i, 2 if i > 201 else 1, [1.0, 5.0], rel.class_cut)
eng.sync()
t_it = (time.perf_counter() - t0) / 200
host = [torch.empty((e.count, 32), dtype=torch.float64).pin_memory() for e in rd.entities]
t0 = time.perf_counter()
for i in range(401, 601):
    eng.step(i, 2, [1.0, 5.0], rel.class_cut)
    eng.sync()
    for h, st in zip(host, eng.ent):
        h.copy_(st.sample[:h.shape[0]], non_blocking=True)
    torch.cuda.synchronize()
t_rb = (time.perf_counter() - t0) / 200
nbytes = sum(h.numel() * 8 for h in host)
print(f"data {source}: set-up {t_setup * 1e3:.0f} ms once (host index build, upload of {rel.data.nnz()} ratings + {len(rel.test_vec.values)} test pairs, plans); "
      f"iteration {t_it * 1e6:.1f} us ({1 / t_it:.0f} sweeps/s, samples stay in HBM); "
      f"iteration + both factor matrices copied to pinned host memory every iteration ({nbytes / 1e6:.2f} MB): {t_rb * 1e6:.1f} us ({1 / t_rb:.0f} sweeps/s)")
eng.close()
