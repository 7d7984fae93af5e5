"""Race soak: two runs of N sweeps of the bench workload (rows on three rotating buffers, hand-overs by polling or events, three
streams, prediction updates beside the rows) must end bit-identical -- a missed dependency between the streams shows up as a
difference sooner or later.
   python tools/soak_determinism.py [N]            the bench's schedule
   python tools/soak_determinism.py [N] rccl       the same with a ONE-rank RCCL communicator in the iteration (ncclAllGather kernels
                                                   on the device between the row launches, the hyperprior's sums through
                                                   bdf_hyper_sums_ranks) and the row kernels still polling for the draws
                                                   (BDF_POLL_WITH_COMM); the run without a communicator agrees to rounding (another order of the hyperprior's sums)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rccl = len(sys.argv) > 2 and sys.argv[2] == "rccl"
if rccl:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 90))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
import bdf_amd as B
from bdf_amd import datasets
outs = []
for rep in range(3 if rccl else 2):
    with_comm = rccl and rep < 2
    for k in ("BDF_FORCE_COMM", "BDF_POLL_WITH_COMM"):
        os.environ.pop(k, None)
        if with_comm:
            os.environ[k] = "1"
    rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
    rel = rd.relations[0]
    eng = B.GibbsEngine(rd, 32, seed=7, device=0)
    test = eng.test_pairs()
    for i in range(1, N + 1):
        eng.step(i, 0 if i < 100 else (1 if i == 100 else 2), [1.0, 5.0], rel.class_cut)
    eng.sync(); torch.cuda.synchronize()
    outs.append((eng.ent[0].sample.cpu().numpy().copy(), eng.ent[1].Lambda.cpu().numpy().copy(), test.stats.cpu().numpy().copy()))
    print(f"run {rep}: native={eng.native} communicator={eng.comm.transport if eng.comm is not None else None} "
          f"unfinished={eng.ctx.rows_unfinished()} rmse={np.sqrt(outs[-1][2][0] / 500000):.6f}", flush=True)
    eng.close()
if rccl:
    # the two runs with the communicator: bit for bit; the run without it adds the hyperprior's sums in another order (the ranks'
    # path cuts them by chunk): the chains agree to rounding
    same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    r0, r2 = np.sqrt(outs[0][2][0] / 500000), np.sqrt(outs[2][2][0] / 500000)
    print(f"held-out RMSE with / without the communicator: {r0:.8f} / {r2:.8f}")
    same = same and abs(r0 - r2) < 1e-4
else:
    same = all(np.array_equal(a, b) for o in outs[1:] for a, b in zip(outs[0], o))
print("bit-identical:", same)
if rccl:
    dist.destroy_process_group()
sys.exit(0 if same else 1)
