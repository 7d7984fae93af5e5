"""Four engines on the same data, one after the other in one process: sweep time and row-kernel times of each.  On fresh side
streams per engine the second one ran 35 % slower (its movies launch 62 instead of 43 us: another mapping of streams onto
hardware queues); the engine therefore reuses the first engine's side streams (engine._SIDE_STREAMS) -- all four equal."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bdf_amd as B
from bdf_amd import datasets
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
def run(label, n=400):
    eng = B.GibbsEngine(rd, 32, seed=1, device=0)
    test = eng.test_pairs()
    for i in range(1, 301):
        eng.sweep(i); test.update(32, eng.factors_of(rel), rel.model.mean_value, 0, [1.0, 5.0], rel.class_cut)
    eng.sync(); torch.cuda.synchronize()
    eng.k1_events = []; eng.k1_event_every = 4
    t0 = time.perf_counter()
    for i in range(301, 301 + n):
        eng.sweep(i); test.update(32, eng.factors_of(rel), rel.model.mean_value, 2, [1.0, 5.0], rel.class_cut)
    eng.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ku = [t.elapsed_us() for (j, t) in eng.k1_events if j == 0]; km = [t.elapsed_us() for (j, t) in eng.k1_events if j == 1]
    ptrs = [hex(eng.ent[j].sample.data_ptr()) for j in (0, 1)] + [hex(eng.ent[j].sample_alt.data_ptr()) for j in (0, 1)]
    print(f"   K1 users {sum(ku)/len(ku):.1f} us, movies {sum(km)/len(km):.1f} us; sample buffers {ptrs}")
    print(f"{label}: {1e6 * dt / n:.1f} us/sweep  native={eng.native} streams: main={eng.ctx.stream.cuda_stream:#x} side={eng.ctx_h.stream.cuda_stream:#x} pred={eng.ctx_p.stream.cuda_stream:#x}")
    eng.close()
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "burn":
    for k in range(int(sys.argv[2])): torch.cuda.Stream(torch.device("cuda", 0))
for k in range(4):
    run(f"engine {k}")
    if mode == "empty":
        torch.cuda.empty_cache()
