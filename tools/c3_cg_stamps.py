"""Where an iteration of the one-launch conjugate-gradient solve (k_cg_resident, configuration C3 with CG forced: F'F 500 x 500, 32 columns,
32 workgroups) spends its time: workgroup 0's clock at six points of every iteration (diagnostic build -DBDF_CG_STAMPS:
tools/ab_k1.sh build cgst "-DBDF_CG_STAMPS"; BDF_LIB_PATH=.../variants/libbdf_cgst.so python3 tools/c3_cg_stamps.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import lib

F = np.random.default_rng(4242).standard_normal((6040, 500))
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rd.entities[0].F = F
eng = B.GibbsEngine(rd, 32, seed=1, device=0, compute_ff_size=0)
for i in range(1, 12):
    eng.sweep(i)
eng.sync()
L = lib()
L.bdf_debug_cg_stamps.argtypes = [C.c_void_p]
acc = []
for rep in range(8):
    eng.sweep(20 + rep)
    eng.sync()
    buf = np.zeros(64 * 8, dtype=np.uint64)
    L.bdf_debug_cg_stamps(buf.ctypes.data_as(C.c_void_p))
    s = buf.reshape(64, 8).astype(np.int64)
    its = [i for i in range(1, 64) if s[i, 0] > 0 and s[i, 5] > 0]
    acc.append(np.array([[(s[i, k + 1] - s[i, k]) / 100.0 for k in range(5)] + [(s[i + 1, 0] - s[i, 5]) / 100.0 if (i + 1) in its else np.nan] for i in its]))
it = eng.ent[0].cg_iters.cpu().numpy()
print(f"CG iterations per column (last sweep): min {it.min()} max {it.max()}; iterations stamped per solve: {[len(a) for a in acc]}")
names = ["load P (agent scope) + matrix instructions", "add the waves' sums, store Z's rows", "first hand-over (drain, arrive, poll)",
         "column step (read Z's column, two block sums, store p)", "second hand-over (drain, arrive, poll)", "loop top (nactive read)"]
allr = np.concatenate(acc, axis=0)
print(f"per iteration, us (workgroup 0, mean over {len(allr)} iterations of {len(acc)} solves; s_memrealtime: 10 ns ticks)")
for k, nm in enumerate(names):
    col = allr[:, k]
    col = col[~np.isnan(col)]
    print(f"  {nm:58s} mean {col.mean():6.2f}  p10 {np.quantile(col, .1):6.2f}  p90 {np.quantile(col, .9):6.2f}")
print(f"  {'iteration':58s} mean {np.nansum(allr, axis=1).mean():6.2f}")
eng.close()
