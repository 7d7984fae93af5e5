"""Steady-state timeline of the two chains of a sweep, from HIP events attached to the kernels' own dispatch packets (no
marker packets, no profiler): begin/end of each row kernel (row stream) and begin of the hyperprior sums / end of the
hyperprior draw (side stream), relative to the first row kernel of the sweep."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import lib, check
from bdf_amd.engine import KernelTimer
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
test = eng.test_pairs()
L = lib()
def us(a, b):
    v = C.c_double(0.0); check(L.bdf_event_elapsed_us(a, b, C.byref(v))); return v.value
marks = []
pool = [KernelTimer() for _ in range(64)]       # created up front: hipEventCreate in the loop makes the host the bottleneck
orig_sample, orig_prior = eng.sample_entity, eng.update_prior
def sample_entity(j):
    t = pool.pop(); check(L.bdf_ctx_time_next_rows(eng.ctx.handle, t.start, t.stop)); marks.append((f"rows{j}", t)); orig_sample(j)
def update_prior(j, sweep=None):
    t = pool.pop(); check(L.bdf_ctx_time_next_hyper(eng.ctx_h.handle, t.start, t.stop)); marks.append((f"hyper{j}", t)); orig_prior(j, sweep)
def step(i, phase):
    eng.sweep(i)
    if not os.environ.get("NO_PREDICT"):
        test.update(32, eng.factors_of(rel), rel.model.mean_value, phase, [1.0, 5.0], rel.class_cut)
class _Proxy:
    """experiment: the row stream with selected waits left out (SKIPW=pred,h0,h1; racy, timing only)"""
    def __init__(self, st): self._st = st; self.n = 0; self.skip = set(os.environ.get("SKIPW", "").split(","))
    def __getattr__(self, k): return getattr(self._st, k)
    def wait_event(self, ev):
        name = ("pred", "h0", "h1")[self.n % 3]; self.n += 1
        if name not in self.skip: self._st.wait_event(ev)
for i in range(1, 301):
    step(i, 0)
if os.environ.get("SKIPW"):
    eng.sync(); torch.cuda.synchronize()
    eng.ctx.stream = _Proxy(eng.ctx.stream)
    for i in range(301, 400):
        step(i, 0)
eng.sync(); torch.cuda.synchronize()
eng.sample_entity, eng.update_prior = sample_entity, update_prior
NS = 12
for k in range(NS):
    step(1000 + k, 2)
eng.sync(); torch.cuda.synchronize()
t0 = marks[4 * 2][1].start       # first row kernel of the third timed sweep
print("times in us relative to the start of rows0 of a steady-state sweep")
prev_end = {}
for name, t in marks[4 * 2: 4 * 7]:
    a, b = us(t0, t.start), us(t0, t.stop)
    print(f"{name:7s} {a:8.1f} -> {b:8.1f}  ({b - a:5.1f})")
span = us(marks[4 * 2][1].start, marks[4 * 7][1].start) / 5
print(f"sweep period {span:.1f} us (timed every launch)")
