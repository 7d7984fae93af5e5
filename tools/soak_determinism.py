"""Race soak: two runs of N sweeps of the bench workload (rows on three rotating buffers, event hand-overs, three streams,
prediction updates beside the rows) must end bit-identical -- a missed dependency between the streams shows up as a
difference sooner or later."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bdf_amd as B
from bdf_amd import datasets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
outs = []
for rep in range(2):
    rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
    rel = rd.relations[0]
    eng = B.GibbsEngine(rd, 32, seed=7, device=0)
    test = eng.test_pairs()
    for i in range(1, N + 1):
        eng.step(i, 0 if i < 100 else (1 if i == 100 else 2), [1.0, 5.0], rel.class_cut)
    eng.sync(); torch.cuda.synchronize()
    outs.append((eng.ent[0].sample.cpu().numpy().copy(), eng.ent[1].Lambda.cpu().numpy().copy(), test.stats.cpu().numpy().copy()))
    print(f"run {rep}: native={eng.native} unfinished={eng.ctx.rows_unfinished()} rmse={np.sqrt(outs[-1][2][0] / 500000):.6f}")
    eng.close()
same = all(np.array_equal(a, b) for a, b in zip(*outs))
print("bit-identical:", same)
sys.exit(0 if same else 1)
