"""Per-wave phase times of K1c (k_rows_col.hip; diagnostic build: tools/ab_k1.sh build stamps "-DBDF_K1_STAMPS").
Run on the GPU box:  BDF_LIB_PATH=.../variants/libbdf_stamps.so python3 tools/col_stamps.py
Per wave: start, end, rounds, observation steps (the longest piece of every round, summed), and the cycles (s_memtime) spent in
normals | accumulation | butterfly sums + slab | prior (incl. waiting for the draw) | factorisation + solves, summed over its rounds."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import lib

D = int(os.environ.get("D", "32"))
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
eng = B.GibbsEngine(rd, D, seed=1, device=0)
for i in range(1, 6):
    eng.sweep(i)
eng.sync()
L = lib()
L.bdf_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
NW = 16384
for j, name in enumerate(("users", "movies")):
    for rep in range(3):
        eng.ctx.set_sweep(10 + j + 2 * rep)
        eng.sample_entity(j)
        eng.sync()
    buf = np.zeros((NW, 16), dtype=np.uint64)
    L.bdf_debug_stamps(eng.ctx.handle, buf.ctypes.data_as(C.c_void_p), NW)
    live = buf[:, 0] > 0
    s = buf[live].astype(np.int64)
    t0, t1 = s[:, 0].min(), s[:, 8].max()
    span = t1 - t0
    print(f"== {name}: {live.sum()} waves, span {span} cycles; rounds/wave {s[:, 6].mean():.2f} (max {s[:, 6].max()}), steps/wave {s[:, 7].mean():.1f} (max {s[:, 7].max()})")
    tot = s[:, 8] - s[:, 0]
    for nm, c in (("normals", 11), ("accumulate", 12), ("sums+slab", 13), ("prior", 14), ("finish", 15)):
        print(f"   {nm:11s} mean {s[:, c].mean():8.0f}  p50 {np.median(s[:, c]):8.0f}  max {s[:, c].max():8d}   share {s[:, c].sum() / tot.sum():.3f}")
    print(f"   wave total  mean {tot.mean():8.0f}  p50 {np.median(tot):8.0f}  max {tot.max():8d}  min {tot.min():8d}")
    print(f"   accumulate cycles per step: {s[:, 12].sum() / max(s[:, 7].sum(), 1):.1f};  finish cycles per round: {s[:, 15].sum() / max(s[:, 6].sum(), 1):.0f};"
          f"  normals per round {s[:, 11].sum() / max(s[:, 6].sum(), 1):.0f};  prior per round {s[:, 14].sum() / max(s[:, 6].sum(), 1):.0f}")
    print("   wave totals deciles", np.round(np.quantile(tot, np.linspace(0, 1, 11))).astype(int))
    for w_ in np.argsort(-tot)[:8]:
        print(f"   slow wave {w_}: total {tot[w_]} rounds {s[w_, 6]} steps {s[w_, 7]} parts {s[w_, 1]} finisher {s[w_, 2]} folds {s[w_, 3]} | normals {s[w_, 11]} acc {s[w_, 12]} sums+slab {s[w_, 13]} prior {s[w_, 14]} finish {s[w_, 15]}")
    multi = s[:, 1] > 0
    if multi.any():
        print(f"   waves with parts of spanning rows: {multi.sum()}; their sums+slab mean {s[multi, 13].mean():.0f} per part {s[multi, 13].sum() / s[multi, 1].sum():.0f}; finisher events {s[:, 2].sum()}")
    fold_only = (s[:, 3] > 0) & ~multi
    if fold_only.any():
        print(f"   waves with butterfly sums only: {fold_only.sum()}; their sums+slab mean {s[fold_only, 13].mean():.0f}; waves with neither: {(~multi & ~(s[:, 3] > 0)).sum()}, sums+slab mean {s[~multi & ~(s[:, 3] > 0), 13].mean():.0f}")
    st = (s[:, 0] - t0) / span
    en = (s[:, 8] - t0) / span
    print("   start deciles ", np.round(np.quantile(st, np.linspace(0, 1, 11)), 2))
    print("   end   deciles ", np.round(np.quantile(en, np.linspace(0, 1, 11)), 2))
    hw = s[:, 9]
    key = (s[:, 10] & 0xf) * (1 << 20) + (hw & 0xfff0)
    ids = np.nonzero(live)[0]
    diffs = []
    for kx in np.unique(key)[:2000]:
        m = np.nonzero(key == kx)[0]
        if len(m) == 2: diffs.append(int(abs(ids[m[1]] - ids[m[0]])))
    vals, cnts = np.unique(diffs, return_counts=True)
    top = np.argsort(-cnts)[:8]
    print("   wave-number distance of the two waves of a SIMD (distance: SIMDs):", {int(vals[t]): int(cnts[t]) for t in top})
    pair_tot = []
    for kx in np.unique(key):
        m = key == kx
        pair_tot.append(tot[m].sum())
    pair_tot = np.array(pair_tot)
    print(f"   per-SIMD sum of its waves' lifetimes: mean {pair_tot.mean():.0f} min {pair_tot.min()} max {pair_tot.max()}; longest wave of a SIMD: mean {np.mean([tot[key == kx].max() for kx in np.unique(key)]):.0f}")
    nw, fin_t, busy = [], [], []
    for kx in np.unique(key):
        m = key == kx
        nw.append(int(m.sum())); fin_t.append((s[m, 8].max() - t0) / span); busy.append(tot[m].sum() / span)
    nw, fin_t, busy = np.array(nw), np.array(fin_t), np.array(busy)
    print(f"   SIMDs seen {len(nw)}; waves per SIMD min {nw.min()} mean {nw.mean():.2f} max {nw.max()}; SIMD finish / span: mean {fin_t.mean():.3f} p10 {np.quantile(fin_t, .1):.3f} p90 {np.quantile(fin_t, .9):.3f};"
          f" wave-time per SIMD / span: mean {busy.mean():.2f} min {busy.min():.2f} max {busy.max():.2f}")
eng.close()
