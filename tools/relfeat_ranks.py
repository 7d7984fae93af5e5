"""Relation-level side information (sample_beta_rel, macau.jl:88-92) and alpha sampling (sample_alpha, macau.jl:86-87) on RANKS
ranks: a 400 x 300 relation (12,000 observations of a planted rank-3 model + 3 observation-level features, CSR or dense), entity
features on the rows as well, D = 8:
   python tools/relfeat_ranks.py [dense|csr]                              one process
   BDF_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... tools/relfeat_ranks.py    two ranks on one GPU
Each rank holds one block of the observations (its rows of the relation's feature matrix, its observations as pairs); the
squared-error sum, F'v and F'F are summed over the ranks in rank order.  Prints one JSON line (rank 0); the chains of the two
runs agree up to the summation order."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bdf_amd as B

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
backend = os.environ.get("BDF_DIST_BACKEND", "nccl")
torch.cuda.set_device(0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)
kind = sys.argv[1] if len(sys.argv) > 1 else "dense"
rng = np.random.default_rng(77)
N1, N2, nnz, R, D = 400, 300, 12_000, 3, 8
key = np.unique(rng.integers(0, N1 * N2, size=int(nnz * 1.1)))[:nnz]
rng.shuffle(key)                                   # COO order unrelated to the rows
ia, ib = key // N2, key % N2
fa, fb = rng.standard_normal((N1, R)), rng.standard_normal((N2, R))
Frel = rng.standard_normal((nnz, 3)) * (rng.random((nnz, 3)) < (0.4 if kind == "csr" else 1.0))
beta_true = np.array([1.0, -0.5, 2.0])
vals = np.sum(fa[ia] * fb[ib], axis=1) + Frel @ beta_true + 0.2 * rng.standard_normal(nnz)
Fent = rng.standard_normal((N1, 5))
rel = B.Relation((np.stack([ia + 1, ib + 1], axis=1), vals), "r", [B.Entity("rows", F=Fent), B.Entity("cols")], dims=[N1, N2])
if kind == "csr":
    import scipy.sparse as sp
    rel.F = sp.csr_matrix(Frel)
else:
    rel.F = Frel
rel.model.alpha_sample = True
B.assignToTest(rel, 600, rng=np.random.default_rng(2))
rd = B.RelationData(rel)
sweeps = int(os.environ.get("RELFEAT_SWEEPS", "12"))
eng = B.GibbsEngine(rd, D, seed=9, shard=(rank, world), chunks=int(os.environ.get("RELFEAT_CHUNKS", "0")))
assert eng.native or os.environ.get("BDF_NO_NATIVE")       # the relation model runs inside the native iteration
n_test = len(rel.test_vec.values)
mine = np.arange(n_test * rank // world, n_test * (rank + 1) // world)
test = eng.test_pairs(subset=mine if world > 1 else None)
alphas = []
for i in range(1, 2 * sweeps + 1):
    eng.step(i, 0 if i <= sweeps else (1 if i == sweeps + 1 else 2), [], rel.class_cut)
    alphas.append(float(eng.rel[0].alpha_dev.item()))
eng.sync()
sse = test.stats[:1].clone().cpu()
if dist is not None:
    dist.all_reduce(sse)
dr = eng.rel[0]
out = {"world": world, "kind": kind, "rmse": float(np.sqrt(sse.item() / n_test)), "alpha": alphas[-1], "alpha_5": alphas[4],
       "beta_rel": [float(x) for x in dr.beta.cpu().numpy()], "sample_norm": float(np.linalg.norm(rd.entities[0].model.sample)),
       "linear_norm": float(np.linalg.norm(dr.linear.cpu().numpy()[:rel.data.nnz()])), "value_std": float(vals.std())}
if rank == 0:
    print(json.dumps(out), flush=True)
eng.close()
if dist is not None:
    dist.destroy_process_group()
