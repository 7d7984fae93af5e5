"""Per-wave phase timeline of K1 (diagnostic build: tools/ab_k1.sh build stamps "-DBDF_K1_STAMPS").
Run on the GPU box:  BDF_LIB_PATH=.../variants/libbdf_stamps.so python3 tools/k1_stamps.py
Stamp slots (s_memtime, shader clock): 0 start, 1 accumulated (split: published), 2 partials summed, 3 prior added,
4 normals drawn, 5 factored, 8 done."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import lib

D = int(os.environ.get("D", "32"))
if os.environ.get("DATA") == "c4":        # a C4-shaped relation (rows x cols x observations from C4_SIZES, default a twentieth of C4): D = 64 by default
    D = int(os.environ.get("D", "64"))
    nr, nc, nz = [int(x) for x in os.environ.get("C4_SIZES", "500000,50000,5000000").split(",")]
    rd = datasets.c4_relation_data(B, nr, nc, nz)
else:
    rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
eng = B.GibbsEngine(rd, D, seed=1, device=0)
if os.environ.get("ITEM"):           # observations per work item / per piece of a split row
    eng.ctx.set_item_size(int(os.environ["ITEM"]))
if os.environ.get("PIECE"):
    eng.ctx.set_piece_size(int(os.environ["PIECE"]))
for i in range(1, 6):
    eng.sweep(i)
eng.sync()
L = lib()
L.bdf_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
NW = 65536 if os.environ.get("DATA") == "c4" else 16384
for j, name in enumerate(("users", "movies")):
    if os.environ.get("DATA") == "c4" and j == 0:
        continue          # (the users' launch has more waves than the stamp buffer holds)
    eng.ctx.set_sweep(10 + j)
    eng.sample_entity(j)
    eng.sync()
    buf = np.zeros((NW, 16), dtype=np.uint64)
    L.bdf_debug_stamps(eng.ctx.handle, buf.ctypes.data_as(C.c_void_p), NW)
    live = buf[:, 0] > 0
    s = buf[live].astype(np.int64)
    t0 = s[:, 0].min()
    end = np.where(s[:, 8] > 0, s[:, 8], s[:, 1])
    print(f"== {name}: {live.sum()} waves, span {end.max() - t0} cycles")
    fin = s[:, 8] > 0
    f = s[fin]
    split_fin = f[:, 2] > 0
    print(f"   finishing waves {fin.sum()} (of them split finishers {split_fin.sum()}), publishing-only {(~fin).sum()}")
    def ph(a, b, sel=None):
        x = f if sel is None else f[sel]
        d = x[:, b] - x[:, a]
        return f"{d.mean():8.0f} (p50 {np.median(d):6.0f} max {d.max():7d})"
    if (s[:, 11:16] > 0).any():       # k_rows4's phase timers (cycles summed over the wave's trips): wait, LDS reads, issue, matrix instructions, flush
        tr = np.maximum(s[:, 6], 1)
        print("   k_rows4 phase timers per wave: trips %.1f; wait %.0f  reads %.0f  issue %.0f  mfma %.0f  flush %.0f (per trip: %.0f %.0f %.0f %.0f)" % (
            s[:, 6].mean(), s[:, 11].mean(), s[:, 12].mean(), s[:, 13].mean(), s[:, 14].mean(), s[:, 15].mean(),
            (s[:, 11] / tr).mean(), (s[:, 12] / tr).mean(), (s[:, 13] / tr).mean(), (s[:, 14] / tr).mean()))
    if (f[:, 6] > 0).any():           # (experiment builds: stamp 6 = the early normals drawn, stamp 7 = the wave got this item)
        sel6 = f[:, 6] > 0
        print("   stamp 6: start->normals", ph(0, 6, sel6), " normals->acc", ph(6, 1, sel6))
    if (f[:, 7] > 0).any():
        sel7 = f[:, 7] > 0
        print("   stamp 7: got item->start", ph(7, 0, sel7))
    print("   start->acc     ", ph(0, 1))
    print("   acc->prior     ", ph(1, 3, ~split_fin) if (~split_fin).any() else "-")
    print("   prior->rng     ", ph(3, 4))
    print("   rng->factor    ", ph(4, 5))
    print("   factor->done   ", ph(5, 8))
    print("   total          ", ph(0, 8))
    # when do waves start / end (deciles of the span)
    span = end.max() - t0
    st = (s[:, 0] - t0) / span
    en = (end - t0) / span
    print("   start deciles ", np.round(np.quantile(st, np.linspace(0, 1, 11)), 2))
    print("   end   deciles ", np.round(np.quantile(en, np.linspace(0, 1, 11)), 2))
    # phase times by position in the launch (wave id deciles): the last waves run on a nearly empty chip
    ids = np.nonzero(live)[0]
    fid = ids[fin]
    for lo in range(0, 10):
        sel = (fid >= np.quantile(fid, lo / 10)) & (fid <= np.quantile(fid, (lo + 1) / 10))
        x = f[sel]
        print(f"   wid decile {lo}: acc {np.mean(x[:,1]-x[:,0]):7.0f} prior {np.mean(x[:,3]-np.maximum(x[:,1],x[:,2])):6.0f} rng {np.mean(x[:,4]-x[:,3]):6.0f} "
              f"factor {np.mean(x[:,5]-x[:,4]):7.0f} bwd {np.mean(x[:,8]-x[:,5]):6.0f} total {np.mean(x[:,8]-x[:,0]):7.0f}")
    # residency: group the waves by the SIMD they ran on (XCC, SE, SH, CU, SIMD from HW_ID) and count how many were
    # live at the same time
    hw = s[:, 9]
    key = (s[:, 10] & 0xf) * (1 << 20) + (hw & 0xfff0)          # xcc | se, sh, cu, pipe, simd
    conc_max, conc_avg, busy = [], [], []
    for kx in np.unique(key):
        m = key == kx
        ev = sorted([(t, 1) for t in s[m, 0]] + [(t, -1) for t in end[m]])
        cur, last, area, mx, first = 0, ev[0][0], 0, 0, ev[0][0]
        for t, dlt in ev:
            area += cur * (t - last); last = t; cur += dlt; mx = max(mx, cur)
        conc_max.append(mx); conc_avg.append(area / max(last - first, 1)); busy.append(last - first)
    print(f"   SIMDs seen {len(conc_max)}; waves live at once per SIMD: max {np.max(conc_max)} mean-of-max {np.mean(conc_max):.2f} "
          f"time-average {np.mean(conc_avg):.2f}; SIMD active span mean {np.mean(busy):.0f} max {np.max(busy)} ticks; items per SIMD {m.size / len(conc_max):.2f}")
    # balance: when does each SIMD finish its last wave (relative to the launch's first start, in units of the span)?
    fin_t, n_w = [], []
    for kx in np.unique(key):
        m = key == kx
        fin_t.append((end[m].max() - t0) / span); n_w.append(int(m.sum()))
    fin_t = np.array(fin_t); n_w = np.array(n_w)
    print(f"   per-SIMD finish time / span: mean {fin_t.mean():.3f} p10 {np.quantile(fin_t, .1):.3f} p50 {np.median(fin_t):.3f} p90 {np.quantile(fin_t, .9):.3f} max {fin_t.max():.3f};"
          f" waves per SIMD min {n_w.min()} mean {n_w.mean():.2f} max {n_w.max()}; corr(finish, waves) {np.corrcoef(fin_t, n_w)[0, 1]:.2f}")
    # per CU
    keyc = key >> 2 if False else (s[:, 10] & 0xf) * (1 << 20) + (hw & 0xff00)
    # the slowest waves
    tot = end - s[:, 0]
    worst = np.argsort(-tot)[:5]
    for w in worst:
        print("   slow wave", int(np.nonzero(live)[0][w]), "stamps", (s[w] - t0)[[0, 1, 2, 3, 4, 5, 8]])
eng.close()
