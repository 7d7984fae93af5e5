"""Where a native iteration's time goes on the bench workload (GPU box, no profiler): pace of 400 iterations
 (a) as the bench runs them (rows of both entities + hyperpriors + prediction update),
 (b) without the prediction update,
 (c) the two row launches alone, back to back (bdf_gibbs_rows_only),
so that (b) - (c) is what the hand-overs between the row launches cost and (a) - (b) what the prediction update costs beside them."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import check, lib

rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1)
test = eng.test_pairs()
eng.warm_device(60.0)
n = 400
def pace(fn):
    for i in range(1, 101):
        fn(i)
    eng.sync()
    t0 = time.perf_counter()
    for i in range(101, 101 + n):
        fn(i)
    eng.sync()
    return 1e6 * (time.perf_counter() - t0) / n
a = pace(lambda i: eng.step(i, 2 if i > 1 else 1, [1.0, 5.0], rel.class_cut))
b = pace(lambda i: eng.sweep(i))
def rows_only(i):
    check(lib().bdf_gibbs_rows_only(eng.gibbs, 0, 1_000_000 + 2 * i))
    check(lib().bdf_gibbs_rows_only(eng.gibbs, 1, 1_000_001 + 2 * i))
c = pace(rows_only)
from bdf_amd.engine import _ptr
eng.sync()
side = []
for st in eng.ent:               # a frozen copy of every entity's rows and outputs of their own: nothing of the chain is touched
    side.append(dict(S=st.sample.clone(), sumU=st.sumU.clone(), UUt=st.UUt.clone(), mu=st.mu.clone(), Lam=st.Lambda.clone(),
                     par=st.params.clone(), pack=st.prior_pack.clone()))
eng.sync()
def rows_beside_hyper(i):        # ... with the hyperprior kernels of both entities on the reserved CUs beside them, no dependence
    rows_only(i)
    eng.ctx_h.set_sweep(3_000_000 + i)
    h = eng.ctx_h.handle
    for st, b in zip(eng.ent, side):
        check(lib().bdf_hyper_sums(h, 32, st.N, _ptr(b["S"]), None, _ptr(b["sumU"]), _ptr(b["UUt"])))
        check(lib().bdf_hyper_sample(h, 32, st.n_real, _ptr(b["sumU"]), _ptr(b["UUt"]), _ptr(st.mu0), st.b0, _ptr(st.WI), st.nu0, st.tag,
                                     _ptr(b["mu"]), _ptr(b["Lam"]), _ptr(b["par"]), _ptr(b["pack"]), None))
e = pace(rows_beside_hyper)
print(f"the two row launches with independent hyperprior kernels beside them on the reserved CUs: {e:.1f} us")
from bdf_amd.engine import KernelTimer
ts = [KernelTimer(), KernelTimer()]
def rows_with_events(i):        # ... each with a completion event on its dispatch, as the iteration attaches one
    for j in (0, 1):
        check(lib().bdf_ctx_time_next_rows(eng.ctx.handle, None, ts[j].stop))
        check(lib().bdf_gibbs_rows_only(eng.gibbs, j, 2_000_000 + 2 * i + j))
d = pace(rows_with_events)
print(f"the two row launches alone with a completion event riding on each dispatch: {d:.1f} us")
print(f"iteration with the prediction update {a:.1f} us; without it {b:.1f} us; the two row launches alone {c:.1f} us "
      f"(hand-overs {b - c:.1f} us, prediction update beside the rows {a - b:.1f} us)")
eng.close()
