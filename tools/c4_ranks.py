"""A C4-shaped relation (default 2M x 200k, 20M observations, D = 64) on 1 or 2 ranks (two ranks on one GPU: the test rig,
BDF_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/c4_ranks.py):
the chain's held-out RMSE, and the time of the hyperprior's sums of the users by themselves -- one rank adds all the rows
(bdf_hyper_sums), R ranks add their own rows and gather D + D^2 partial sums (bdf_hyper_sums_ranks).  One JSON line (rank 0)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd._lib import check, lib
from bdf_amd.engine import _ptr

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
backend = os.environ.get("BDF_DIST_BACKEND", "nccl")
torch.cuda.set_device(0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)
rows, cols, nnz = [int(x) for x in os.environ.get("C4_SIZE", "2000000,200000,20000000").split(",")]
rd = datasets.c4_relation_data(B, rows, cols, nnz)
rel = rd.relations[0]
D, sweeps = 64, int(os.environ.get("C4_SWEEPS", "3"))
eng = B.GibbsEngine(rd, D, seed=5, shard=(rank, world))
n_test = len(rel.test_vec.values)
mine = np.arange(n_test * rank // world, n_test * (rank + 1) // world)
test = eng.test_pairs(subset=mine if world > 1 else None)
for i in range(1, 2 * sweeps + 1):
    eng.step(i, 0 if i <= sweeps else (1 if i == sweeps + 1 else 2), [1.0, 5.0], rel.class_cut)
eng.sync()
sse = test.stats[:1].clone().cpu()
if dist is not None:
    dist.all_reduce(sse)
out = {"world": world, "rows": rows, "rmse": float(np.sqrt(sse.item() / n_test)), "chunks": eng.layouts[0].chunks}
st = eng.ent[0]
h = eng.ctx_h
torch.cuda.synchronize()
for rep in range(2):
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    n = 10
    for k in range(n):
        if world > 1:
            check(lib().bdf_hyper_sums_ranks(h.handle, eng.comm.handle, D, st.N, st.layout.chunks, _ptr(st.sample), None, _ptr(st.sumU), _ptr(st.UUt)))
        else:
            check(lib().bdf_hyper_sums(h.handle, D, st.N, _ptr(st.sample), None, _ptr(st.sumU), _ptr(st.UUt)))
    h.sync()
    tb = torch.tensor([(time.perf_counter() - t0) / n * 1e3], dtype=torch.float64)
if dist is not None:
    dist.all_reduce(tb, op=dist.ReduceOp.MAX)
out["users_hyper_sums_ms"] = round(float(tb.item()), 3)
out["sumU_norm"] = float(st.sumU.norm().item())
out["UUt_norm"] = float(st.UUt.norm().item())
if rank == 0:
    print(json.dumps(out), flush=True)
eng.close()
