"""Where do the row kernel's workgroups land (diagnostic build -DBDF_K1_STAMPS; GPU box)?  Prints, for the users' launch, the
(XCC, SE, CU) of consecutive workgroups and how periodic the placement is: the period with which workgroups share a CU."""
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
NW = 16384
for rep in range(2):
    eng.ctx.set_sweep(10 + rep)
    eng.sample_entity(0)
    eng.sync()
    buf = np.zeros((NW, 16), dtype=np.uint64)
    L.bdf_debug_stamps(eng.ctx.handle, buf.ctypes.data_as(C.c_void_p), NW)
    live = np.nonzero(buf[:, 0] > 0)[0]
    nw = live.max() + 1
    hw = buf[:nw, 9].astype(np.int64); xcc = buf[:nw, 10].astype(np.int64) & 0xf
    # HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx9 layout)
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    wg = cuid[0:nw - nw % 4:4]
    print(f"rep {rep}: {nw} waves, {len(np.unique(cuid[buf[:nw,0] > 0]))} CUs seen; SIMD of wave 0..7 of the launch: {simd[:8].tolist()}")
    print("   workgroup -> (xcc,se,cu) first 40:", [(int(xcc[4*b]), int(se[4*b]), int(cu[4*b])) for b in range(40)])
    start = buf[:nw, 0].astype(np.int64)
    for P in (8, 31, 32, 62, 64, 124, 128, 248, 256, 496, 512):
        same = np.mean(wg[:-P] == wg[P:]) if len(wg) > P else 0
        print(f"   period {P:4d}: share of workgroups b with CU(b) == CU(b+{P}): {same:.3f}")
    # workgroups per CU and the order index of a CU's workgroups
    first = {}
    for b, c in enumerate(wg):
        first.setdefault(int(c), []).append(b)
    gaps = np.concatenate([np.diff(v) for v in first.values() if len(v) > 1])
    vals, cnt = np.unique(gaps, return_counts=True)
    top = np.argsort(-cnt)[:8]
    print("   most common gaps between a CU's consecutive workgroups:", [(int(vals[i]), int(cnt[i])) for i in top])
eng.close()
