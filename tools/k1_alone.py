"""the row kernel alone: N back-to-back launches alternating the two entities on one stream, wall clock per launch (GPU box)"""
import os, sys, time
NATIVE = os.environ.get("K1_NATIVE", "1") != "0"       # the native engine's row launch (bdf_gibbs_rows_only) or the step-by-step one
if not NATIVE:
    os.environ["BDF_NO_NATIVE"] = "1"; os.environ["BDF_NO_OVERLAP"] = "1"
os.environ.setdefault("BDF_RESERVE_CUS", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B
from bdf_amd import datasets
D = int(os.environ.get("D", "32"))
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
eng = B.GibbsEngine(rd, D, seed=1, device=0)
if os.environ.get("ITEM"):           # observations per work item (pieces of split rows: two thirds of it)
    eng.ctx.set_item_size(int(os.environ["ITEM"]))
if os.environ.get("PIECE"):
    eng.ctx.set_piece_size(int(os.environ["PIECE"]))
for i in range(1, 30):
    eng.sweep(i)
eng.sync()
from bdf_amd._lib import check, lib
def launch(j, sweep):
    if eng.native:
        check(lib().bdf_gibbs_rows_only(eng.gibbs, j, sweep))
    else:
        eng.ctx.set_sweep(sweep)
        eng.sample_entity(j)
best = 1e9
for rep in range(5):
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        launch(i % 2, 100 + i)
    eng.sync()
    best = min(best, (time.perf_counter() - t0) / n)
print(f"K1 alone: {best * 1e6:.2f} us per launch (mean of users' and movies', D={D}, reserve {os.environ['BDF_RESERVE_CUS']})")
for j, name in enumerate(("users", "movies")):
    bj = 1e9
    for rep in range(5):
        n = 300
        t0 = time.perf_counter()
        for i in range(n):
            launch(j, 1000 + i)
        eng.sync()
        bj = min(bj, (time.perf_counter() - t0) / n)
    print(f"   {name}' launches only: {bj * 1e6:.2f} us")
# the test prediction update alone (two kernels: k_predict_runs + k_predict_final), back to back on its stream
test = eng.test_pairs()
r = rd.relations[0]
facs = eng.factors_of(r)
bp = 1e9
for rep in range(5):
    n = 300
    eng.sync()
    t0 = time.perf_counter()
    for i in range(n):
        test.update(D, facs, r.model.mean_value, 2, [1.0, 5.0], 2.5)
    eng.sync()
    bp = min(bp, (time.perf_counter() - t0) / n)
print(f"prediction update alone: {bp * 1e6:.2f} us per update ({test.n} pairs)")
eng.close()
