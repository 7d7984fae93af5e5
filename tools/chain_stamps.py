"""Phase stamps of the hyperprior chain INSIDE the iteration (diagnostic build: tools/ab_k1.sh build hst "-DBDF_HYPER_STAMPS";
run with BDF_LIB_PATH=.../variants/libbdf_hst.so).  MovieLens D = 32, the bench's engine; after every few iterations the last
chain's stamps (100 MHz ticks) are read back."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import lib
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
test = eng.test_pairs()
for i in range(1, 200): eng.sweep(i)
eng.sync()
L = C.CDLL(os.environ["BDF_LIB_PATH"])
names = ["final WG start", "sums+draws arrived", "nw: stage 2 sums + muN", "nw: assemble W", "nw: factor | q", "nw: Z solve | mean", "nw: Z Z'", "nw: (ref mean)", "nw: pack"]
acc = []
for rep in range(40):
    for i in range(200 + 7 * rep, 200 + 7 * rep + 7 + (rep & 1)): eng.sweep(i)      # (odd / even counts: both entities' chains get sampled)
    eng.sync()
    st = (C.c_ulonglong * 16)()
    L.bdf_debug_hyper_stamps(st)
    s = np.array(list(st), dtype=np.int64)
    seq = np.array([s[8], s[9], s[0], s[1], s[2], s[3], s[4], s[5], s[6]])
    acc.append(np.concatenate([np.diff(seq), [s[11] - s[10], s[13] - s[12], s[10] - s[8], s[12] - s[8], s[13] - s[8]]]))
a = np.array(acc) / 100.0
m = a.mean(axis=0)
lab = ["wait for sums+draws", "entry nw_draw", "stage 2 sums + muN", "assemble W", "factor | q", "Z solve | mean", "Z Z'", "(ref mean)", "pack",
       "partial WG 0 duration", "last partial WG duration", "partial WG 0 start - final WG start", "last partial WG start - final start", "last partial WG end - final start"]
for l, v, sd in zip(lab, m, a.std(axis=0)): print(f"{l:42s} {v:7.2f} us  (sd {sd:.2f})")
print(f"final WG start -> pack written: {(a[:, :9].sum(axis=1)).mean():.2f} us")
eng.close()
