"""Device-side timeline of the row launches inside the native iteration, without a profiler (diagnostic build:
tools/ab_k1.sh build spans "-DBDF_K1_SPANS"; GPU box: BDF_LIB_PATH=.../variants/libbdf_spans.so python3 tools/k1_spans.py).
Every launch records the shader-clock time its first wave started and its last wave ended, and how long its waves polled for
the hyperprior draw (longest wave; sum over waves).  Prints per entity: launch duration, the gap to the next launch's first
wave, the waits."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import lib

rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1)
test = eng.test_pairs()
eng.warm_device(60.0)
import time
n = 300
alone = len(sys.argv) > 1 and sys.argv[1] == "alone"       # the two row launches back to back, nothing beside them
from bdf_amd._lib import check
def it(i, phase):
    if alone:
        check(lib().bdf_gibbs_rows_only(eng.gibbs, 0, 1_000_000 + 2 * i))
        check(lib().bdf_gibbs_rows_only(eng.gibbs, 1, 1_000_001 + 2 * i))
    else:
        eng.step(i, phase, [1.0, 5.0], rel.class_cut)
for i in range(1, 101):
    it(i, 0)
eng.sync()
t0 = time.perf_counter()
for i in range(101, 101 + n):
    it(i, 2 if i > 101 else 1)
eng.sync()
pace = 1e6 * (time.perf_counter() - t0) / n
L = lib()
buf = np.zeros((1024, 8192, 3), dtype=np.uint64)
cnt = C.c_ulonglong()
L.bdf_debug_spans.argtypes = [C.c_void_p, C.c_void_p]
L.bdf_debug_spans(buf.ctypes.data_as(C.c_void_p), C.byref(cnt))
total = cnt.value
last = [(k % 1024) for k in range(total - 2 * 200, total)]           # the last 200 iterations' launches, in order
recs = []
for k in last:
    w = buf[k].astype(np.int64)
    live = (w[:, 1] > 0) & (w[:, 0] > 0)
    odd = int(((w[:, 1] > 0) != (w[:, 0] > 0)).sum())
    recs.append((w[live, 0].min(), w[live, 1].max(), w[live, 2].max(), w[live, 2].sum(), int(live.sum()),
                 np.sort(w[live, 0])[int(0.9 * live.sum())], np.percentile(w[live, 1], 50), odd))
r = np.array(recs, dtype=np.float64)
clk = (r[-2, 0] - r[0, 0]) / 199 / pace                               # ticks per microsecond, from the wall-clock pace
print("waves with only one of the two stamps, per launch:", r[:, 7].mean())
dur = (r[:, 1] - r[:, 0]) / clk
gap = (r[1:, 0] - r[:-1, 1]) / clk
print(f"iteration pace {pace:.1f} us by the wall clock ({clk:.1f} ticks of s_memtime per us)")
for e, name in ((0, "users"), (1, "movies")):
    d, g = dur[e::2], gap[e::2]
    w, ws = r[e::2, 2] / clk, r[e::2, 3] / clk
    s90 = (r[e::2, 5] - r[e::2, 0]) / clk
    e50 = (r[e::2, 6] - r[e::2, 0]) / clk
    print(f"  {name:6s} launch ({int(r[e, 4])} waves): first wave start -> last wave end {d.mean():6.2f} us (p10 {np.percentile(d, 10):.2f} p90 {np.percentile(d, 90):.2f}); "
          f"90% of the waves started by {s90.mean():5.2f} us, half ended by {e50.mean():5.2f} us; "
          f"gap to the next launch's first wave {g.mean():5.2f} us (p10 {np.percentile(g, 10):.2f} p90 {np.percentile(g, 90):.2f}); "
          f"longest wait for the prior {w.mean():5.2f} us (max {w.max():.2f}), summed over waves {ws.mean():8.1f} us")
eng.close()
