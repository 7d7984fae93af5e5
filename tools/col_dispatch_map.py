"""where the dispatcher puts the single-wave workgroups of a K1c launch (diagnostic build -DBDF_K1_STAMPS): wave number ->
(XCC, SE, SH, CU, SIMD) from HW_ID / XCC_ID, with and without the row stream's CU mask (GPU box)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import lib
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
for i in range(1, 4):
    eng.sweep(i)
eng.sync()
L = lib()
L.bdf_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
eng.ctx.set_sweep(10)
eng.sample_entity(0)
eng.sync()
NW = 4096
buf = np.zeros((NW, 16), dtype=np.uint64)
L.bdf_debug_stamps(eng.ctx.handle, buf.ctypes.data_as(C.c_void_p), NW)
live = buf[:, 0] > 0
n = int(live.sum())
hw, xcc = buf[:n, 9].astype(np.int64), buf[:n, 10].astype(np.int64) & 0xf
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
print("reserve", os.environ.get("BDF_RESERVE_CUS"), "waves", n)
for w in list(range(0, 72)) + list(range(880, 912)) + list(range(1016, 1040)):
    if w < n:
        print(w, "xcc", xcc[w], "se", se[w], "sh", sh[w], "cu", cu[w], "simd", simd[w])
key = xcc * 4096 + se * 512 + sh * 256 + cu * 4 + simd
first = {}
gens = np.zeros(n, dtype=int)
for w in range(n):
    gens[w] = first.setdefault(int(key[w]), []).__len__()
    first[int(key[w])].append(w)
print("SIMDs used", len(first), "waves per SIMD histogram", np.bincount([len(v) for v in first.values()]))
d = [v[1] - v[0] for v in first.values() if len(v) > 1]
vals, cnt = np.unique(d, return_counts=True)
print("distance gen0 -> gen1:", dict(zip(vals.tolist(), cnt.tolist())))
cus = {}
for w in range(n):
    cus.setdefault((int(xcc[w]), int(se[w]), int(sh[w])), set()).add(int(cu[w]))
print("CUs seen per (xcc, se, sh):", {k: sorted(v) for k, v in sorted(cus.items())})
# first wave number of generation 1 per (xcc, se)
g1 = {}
for v in first.values():
    if len(v) > 1:
        w = v[1]
        k = (int(xcc[w]), int(se[w]))
        g1[k] = min(g1.get(k, 10 ** 9), w)
print("first second-generation wave per (xcc, se):", dict(sorted(g1.items())))
eng.close()
