"""A 20-iteration region as bench.py times it under the driver's command (set-up warm-up, 5 warm-up iterations, full
synchronisation, 20 iterations, full synchronisation), seen from the device without a profiler (diagnostic build
-DBDF_K1_SPANS, see tools/k1_spans.py): per row launch of the region its duration (first wave start -> last wave end) and the
gap to the next launch; the host's wall clock around the region beside it."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import lib

rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1)
test = eng.test_pairs()
eng.warm_device(60.0)
L = lib()
L.bdf_debug_spans.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros((1024, 8192, 3), dtype=np.uint64)
cnt = C.c_ulonglong()
it = 0
for rep in range(3):
    for i in range(5):
        it += 1
        eng.step(it, 0, [1.0, 5.0], rel.class_cut)
    eng.sync(); torch.cuda.synchronize()
    L.bdf_debug_spans(buf.ctypes.data_as(C.c_void_p), C.byref(cnt))
    first = cnt.value
    t0 = time.perf_counter()
    for k in range(20):
        it += 1
        eng.step(it, 1 if k == 0 else 2, [1.0, 5.0], rel.class_cut)
    torch.cuda.synchronize()
    wall = 1e6 * (time.perf_counter() - t0)
    eng.sync()
    L.bdf_debug_spans(buf.ctypes.data_as(C.c_void_p), C.byref(cnt))
    recs = []
    for k in range(first, cnt.value):
        w = buf[k % 1024].astype(np.int64)
        live = (w[:, 1] > 0) & (w[:, 0] > 0)
        recs.append((w[live, 0].min(), w[live, 1].max()))
        buf[k % 1024] = 0
    r = np.array(recs, dtype=np.float64) / 100.0          # s_memrealtime: 100 ticks per us
    dur = r[:, 1] - r[:, 0]
    gap = r[1:, 0] - r[:-1, 1]
    print(f"region {rep}: wall {wall:.0f} us for 20 iterations ({wall / 20:.1f} per iteration); first row launch starts -> last row launch ends "
          f"{r[-1, 1] - r[0, 0]:.0f} us; {len(recs)} launches")
    print("   durations:", " ".join(f"{d:.1f}" for d in dur))
    print("   gaps:     ", " ".join(f"{g:.1f}" for g in gap))
eng.close()
