"""Determinism of the row kernel under load (GPU box): the same launch N times, outputs compared bit for bit with the first.
A hazard the compiler cannot see (inline-asm VALU / LDS instructions next to matrix instructions) shows up as a few rows that
differ on some launches.  Usage: k1_repeat_check.py [D] [rows] [cols] [nnz] [launches]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import check, lib
from bdf_amd.engine import GibbsEngine, _ptr
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, cols, nnz = (int(sys.argv[k]) if len(sys.argv) > k else v for k, v in ((2, 1_000_000), (3, 100_000), (4, 20_000_000)))
N = int(sys.argv[5]) if len(sys.argv) > 5 else 30
rd = datasets.c4_relation_data(B, rows, cols, nnz)
eng = GibbsEngine(rd, D, seed=5)
for i in range(1, 4):
    eng.sweep(i)
eng.sync()
ctx = eng.ctx
bad_total = 0
for j in (0, 1):
    st, terms = eng.ent[j], eng._terms(j)
    ref = None
    for it in range(N):
        out = ctx.zeros(st.N, D)
        ctx.set_sweep(77)
        check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, _ptr(st.mu), 0, _ptr(st.Lambda), st.tag, 0, 1, _ptr(out), None))
        ctx.sync()
        if ref is None:
            ref = out
            continue
        diff = (out != ref).any(dim=1)
        nb = int(diff.sum().item())
        if nb:
            idx = torch.nonzero(diff)[:4, 0].tolist()
            print(f"entity {j} launch {it}: {nb} rows differ from launch 0, e.g. {idx}; max |diff| {float((out - ref).abs().max()):.3e}")
            bad_total += nb
    print(f"entity {j}: {N} launches of {st.N} rows compared, finite: {bool(torch.isfinite(ref).all())}")
print("rows that differed:", bad_total)
eng.close()
