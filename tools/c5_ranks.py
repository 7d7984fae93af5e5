"""Configuration C5's structure (3-mode tensor + matrix sharing an entity with binary sparse features, CG) on RANKS ranks:
   python tools/c5_ranks.py                                   one process
   BDF_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... tools/c5_ranks.py    two ranks on one GPU (test rig)
prints one JSON line (rank 0): held-out RMSE, norms of A's sample (original row order) and of beta -- the chains of the two
runs agree up to the summation order of the hyperprior's sums."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bdf_amd as B
from bdf_amd import datasets

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
backend = os.environ.get("BDF_DIST_BACKEND", "nccl")
torch.cuda.set_device(0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)
sizes = dict(nA=3000, nB=16, nC=60, nT=40, n1=120_000, n2=30_000, n_feat=400, feat_per_row=8) if len(sys.argv) < 2 else {}
rd, info = datasets.c5_relation_data(B, **sizes)
rel = rd.relations[0]
D, sweeps = 32, int(os.environ.get("C5_SWEEPS", "8"))
eng = B.GibbsEngine(rd, D, seed=3, compute_ff_size=0, shard=(rank, world), chunks=int(os.environ.get("C5_CHUNKS", "0")))
n_test = len(rel.test_vec.values)
mine = np.arange(n_test * rank // world, n_test * (rank + 1) // world)
test = eng.test_pairs(subset=mine if world > 1 else None)
for i in range(1, 2 * sweeps + 1):
    eng.step(i, 0 if i <= sweeps else (1 if i == sweeps + 1 else 2), [], rel.class_cut)
eng.sync()
sse = test.stats[:1].clone().cpu()
if dist is not None:
    dist.all_reduce(sse)
A = rd.entities[0]
out = {"world": world, "rmse": float(np.sqrt(sse.item() / n_test)), "sample_norm": float(np.linalg.norm(A.model.sample)),
       "beta_norm": float(np.linalg.norm(A.model.beta)), "lambda_beta": float(eng.ent[0].lambda_beta.item()),
       "cg_iters": int(eng.ent[0].cg_iters.max().item()), "value_std": info["value_std"]}
# the beta update by itself, after the chain: with several ranks every rank solves ceil(D / world) of the D conjugate-gradient
# columns (bdf_sample_beta_ranks) -- its time per call, the slowest rank's
import time
from bdf_amd._lib import check, lib
from bdf_amd.engine import _ptr
st, en = eng.ent[0], A
eng.sync(); torch.cuda.synchronize()
if dist is not None:
    dist.barrier()
t0 = time.perf_counter()
for k in range(3):
    eng.ctx.set_sweep(1000 + k)
    check(lib().bdf_sample_beta_ranks(eng.ctx.handle, eng.comm.handle if eng.comm is not None else None, st.F.handle, D, _ptr(st.sample),
                                      _ptr(st.mu), _ptr(st.Lambda), _ptr(st.lambda_beta), 0, float("nan"), 0, 1, en.nu, en.mu, st.tag,
                                      _ptr(st.beta), None, _ptr(st.cg_iters)))
eng.ctx.sync()
tb = torch.tensor([(time.perf_counter() - t0) / 3 * 1e3], dtype=torch.float64)
if dist is not None:
    dist.all_reduce(tb, op=dist.ReduceOp.MAX)
out["beta_update_ms"] = round(float(tb.item()), 3)
out["beta_columns_per_rank"] = -(-D // world)
if rank == 0:
    print(json.dumps(out), flush=True)
eng.close()
if dist is not None:
    dist.destroy_process_group()
