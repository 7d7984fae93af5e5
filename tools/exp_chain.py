"""Sensitivity of the sweep time to the hyperprior chain: the bench step with the full chain, with the sums only (no draw), and
with no hyperprior kernels at all (timing only: the prior then stays what it was).  Round 1: 131 / 124 / 126 us -- the chain is
not what bounds the sweep."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bdf_amd as B
from bdf_amd import datasets
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
test = eng.test_pairs()
def step(i, phase):
    eng.sweep(i)
    test.update(32, eng.factors_of(rel), rel.model.mean_value, phase, [1.0, 5.0], rel.class_cut)
for i in range(1, 501): step(i, 0)
eng.sync(); torch.cuda.synchronize()
def timeit(label, n=300):
    t0 = time.perf_counter()
    for k in range(n): step(1000 + k, 2)
    eng.sync(); torch.cuda.synchronize()
    print(f"{label}: {1e6 * (time.perf_counter() - t0) / n:.1f} us/sweep")
timeit("full")
orig = eng.update_prior
mode = {"m": 0}
from bdf_amd._lib import lib, check
from bdf_amd.engine import _ptr
def upd(j, sweep=None):
    st = eng.ent[j]
    if mode["m"] == 1:      # sums only
        check(lib().bdf_hyper_sums(eng.ctx_h.handle, eng.D, st.N, _ptr(st.sample), None, _ptr(st.sumU), _ptr(st.UUt)))
    elif mode["m"] == 2:
        pass
eng.update_prior = upd
mode["m"] = 1; timeit("sums only (no draw)")
mode["m"] = 2; timeit("no hyper kernels at all")
eng.update_prior = orig; timeit("full again")
