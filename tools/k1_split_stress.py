"""Stress of the split-row protocol (GPU box): on a C4-shaped relation at D = 64 the row launches of tests' _shards_items_map
(one launch, two shards, item size 64, the row-system dump) are repeated; after every launch group the arrival counters must
all be back at zero (bdf_rows_unfinished) and the samples equal to the first round's, bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import check, lib
from bdf_amd.engine import GibbsEngine, _ptr
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rd = datasets.c4_relation_data(B, 1_000_000, 100_000, 20_000_000)
rel = rd.relations[0]
eng = GibbsEngine(rd, D, seed=5)
eng.ctx.set_gather(2)
test = eng.test_pairs()
for i in range(1, 8):
    eng.step(i, 0, [1.0, 5.0], rel.class_cut)
eng.sync()
ctx = eng.ctx
ref = {}
bad = 0
for rnd in range(rounds):
    for j in (1, 0):
        st, terms = eng.ent[j], eng._terms(j)
        ctx.set_sweep(9)
        def rows(shards, tag):
            global bad
            out = ctx.zeros(st.N, D)
            for s in range(shards):
                check(lib().bdf_sample_rows(ctx.handle, D, st.N, 1, terms, _ptr(st.mu), 0, _ptr(st.Lambda), st.tag, s, shards, _ptr(out), None))
            ctx.sync()
            u = ctx.rows_unfinished()
            key = (j, tag)
            if key not in ref:
                ref[key] = out
            nd = int((out != ref[key]).any(dim=1).sum().item())
            if u or nd:
                bad += 1
                print(f"round {rnd} entity {j} {tag}: unfinished {u}, rows differing from round 0: {nd}", flush=True)
        rows(1, "one"); rows(2, "two shards")
        ctx.set_item_size(64); rows(1, "item 64"); ctx.set_item_size(192)
        if j == 1:
            P_t, b_t = ctx.zeros(st.N, D, D), ctx.zeros(st.N, D)
            check(lib().bdf_row_system(ctx.handle, D, st.N, 1, terms, _ptr(st.mu), 0, _ptr(st.Lambda), _ptr(P_t), _ptr(b_t)))
            ctx.sync()
            u = ctx.rows_unfinished()
            if u:
                bad += 1
                print(f"round {rnd} entity {j} row_system: unfinished {u}", flush=True)
            del P_t, b_t
print(f"{rounds} rounds, failures: {bad}")
eng.close()
