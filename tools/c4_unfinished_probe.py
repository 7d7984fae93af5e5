"""Where do split rows stay unfinished (GPU box)?  The C4-shaped test's scenario with bdf_rows_unfinished after every step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd.engine import GibbsEngine
mode = sys.argv[1] if len(sys.argv) > 1 else "step"
rd = datasets.c4_relation_data(B, 1_000_000, 100_000, 20_000_000)
rel = rd.relations[0]
eng = GibbsEngine(rd, 64, seed=5)
eng.ctx.set_gather(2)
test = eng.test_pairs()
print("native", eng.native, "mode", mode, flush=True)
for i in range(1, 21):
    if mode == "step":
        eng.step(i, 0 if i <= 10 else (1 if i == 11 else 2), [1.0, 5.0], rel.class_cut)
    else:
        eng.sweep(i)           # no prediction update
    eng.sync()
    torch.cuda.synchronize()
    u = eng.ctx.rows_unfinished()
    if u:
        print(f"after iteration {i}: {u} split rows unfinished", flush=True)
print("final unfinished", eng.ctx.rows_unfinished())
eng.close()
