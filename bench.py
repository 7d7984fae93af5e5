#!/usr/bin/env python3
"""bench.py -- Gibbs sweeps/sec (both entities) + test RMSE, BPMF on MovieLens-1M, D=32, fp64 (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one Gibbs iteration of src/macau.jl:80 on the device: latent rows of users, hyperprior of users, latent rows of
movies, hyperprior of movies, and the test-set prediction update of macau.jl:142-184 (kept inside the timed region; the
reporting-only metrics are not).  W warm-up steps are the burn-in; the K timed steps are the posterior samples, so the
RMSE printed is that of the posterior-mean prediction after W + K iterations on the 500,000 held-out ratings.

N > 1 (weak scaling): MovieLens is a ~130 us sweep, far too small to split, so the N-GPU workload is N MovieLens-sized
units -- the rating matrix stacked over N disjoint user blocks that rate the same movies (N x 6040 users, N x 500,209
training ratings, N x 500,000 held-out ratings; datasets.replicate_users).  One process per GPU; every rank holds the
whole relation and a replica of both factors, samples its share of the rows of each entity (rank, rank + N, ... of the
degree order), and the ranks exchange the freshly sampled rows by an RCCL all-gather after each half-sweep; the test
ratings of user block r are predicted by rank r.  value = N units x sweeps/s, so that N = 1 is exactly the BASELINE
configuration and ideal scaling is N x its value.  (--replicas R runs the R-unit workload on fewer GPUs.)

Prints one JSON line (rank 0).  roofline: K1 (k_sample_rows) algorithmic bytes per launch over its mean
launch duration from HIP events attached to the kernel dispatches (on the launch stream) inside the timed region.  cpu_baseline: the CPU oracle
(a C port of the reference algorithm, OpenMP over rows like the reference's latent_pids workers) timed on this box.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(rd, D, seed, budget_s=12.0):
    """Sweeps/s of the CPU oracle on the same training set (rows of both entities + hyperprior draws)."""
    import numpy as np
    from oracle import oracle as O
    r = rd.relations[0]
    N = list(r.data.dims)
    idx = O.index_build(r.data.ids, N)
    S = [np.zeros((N[0], D)), np.zeros((N[1], D))]
    mu = [np.zeros(D), np.zeros(D)]
    Lam = [5.0 * np.eye(D), 5.0 * np.eye(D)]
    mean = r.data.valueMean()
    nthreads = O.num_threads()

    def sweep(it, nt):
        for j in (0, 1):
            t = O.Term(r.data.ids, r.data.values, N, j, r.model.alpha, mean, [None if k == j else S[k] for k in (0, 1)], index=idx)
            S[j] = O.sample_rows(D, N[j], [t], mu[j], Lam[j], seed, it, j + 1, nthreads=nt)
            mu_N, beta_N, T_N, nu_N = O.hyper_params(S[j], np.zeros(D), 2.0, np.eye(D), float(D))
            mu[j], Lam[j] = O.hyper_draw(mu_N, beta_N, T_N, nu_N, seed, it, j + 1)

    sweep(1, nthreads)                      # untimed: first-touch, and a non-zero factor state
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < budget_s and n < 200):
        sweep(2 + n, nthreads)
        n += 1
    multi = n / (time.time() - t0)
    t0 = time.time()
    sweep(1000, 1)
    sweep(1001, 1)
    single = 2 / (time.time() - t0)
    return {"value": round(multi, 4), "unit": "sweeps/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} full Gibbs sweeps (rows of both entities + hyperpriors) of the same MovieLens D={D} training set, "
                      f"oracle/bdf_oracle.c with OpenMP over rows on {nthreads} threads; single thread: {single:.4f} sweeps/s",
            "value_1thread": round(single, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--num-latent", type=int, default=32)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--k1-event-every", type=int, default=8,
                    help="time the K1 launches of every n-th step (HIP events attached to the kernel dispatch)")
    ap.add_argument("--no-predict", action="store_true", help="leave the test-set prediction update out of the step")
    ap.add_argument("--replicas", type=int, default=0, help="user blocks of the workload (default: one per GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import bdf_amd as B
    from bdf_amd import datasets

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test rig: BDF_DIST_BACKEND=gloo runs all ranks on GPU 0 with the collectives staged through the host (RCCL needs one
    # GPU per rank); it checks the N > 1 logic on a 1-GPU box, its timings mean nothing
    backend = os.environ.get("BDF_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if backend == "gloo" else "cuda"

    D = args.num_latent
    replicas = args.replicas if args.replicas > 0 else world
    rd, source = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5, replicas=replicas)
    rel = rd.relations[0]
    eng = B.GibbsEngine(rd, D, seed=args.seed, device=local_rank, shard=(rank, world))
    if world > 1:
        # rank r predicts the held-out ratings of the user blocks r, r + world, ...
        from bdf_amd.engine import DevicePairs
        tv = rel.test_vec
        tids = np.asarray(tv.ids).reshape(-1, 2)
        block = (tids[:, 0] - 1) // (rel.data.dims[0] // replicas)
        mine = np.nonzero(block % world == rank)[0]
        test = DevicePairs(eng.ctx_p, tids[mine], np.asarray(tv.values)[mine])
    else:
        test = eng.test_pairs()
    n_test_total = len(np.asarray(rel.test_vec.values))
    clamp = [1.0, 5.0]

    def step(i, phase):
        eng.sweep(i)
        if not args.no_predict:
            test.update(D, eng.factors_of(rel), rel.model.mean_value, phase, clamp, rel.class_cut)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(1, args.warmup + 1):
        step(i, 0)
    eng.sync()
    fence()
    eng.k1_events = []
    eng.k1_event_every = max(1, min(args.k1_event_every, args.steps // 4))     # a short run still times a few launches
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + 1 + k, 1 if k == 0 else 2)
    fence()
    elapsed = time.perf_counter() - t0
    eng.sync()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # K1 roofline from the events recorded inside the timed region
    k1_ms = sum(t.elapsed_us() for (_, t) in eng.k1_events) / 1e3
    k1_bytes = sum(eng.k1_algorithmic_bytes(j) for (j, _) in eng.k1_events) / max(world, 1)
    n_launch = max(len(eng.k1_events), 1)
    achieved = (k1_bytes / 1e9) / (k1_ms / 1e3) if k1_ms > 0 else 0.0
    eng.k1_events = None

    # HBM traffic of one K1 launch from the PMC passes of the same workload (tools/profile_round.sh: FETCH_SIZE and
    # WRITE_SIZE in separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), if recorded
    traffic = None
    tj = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tj) and world == 1 and replicas == 1:
        try:
            traffic = json.load(open(tj))["k1_traffic_bytes_per_launch"]["hbm_bytes_fetch_doubled"]
        except (KeyError, ValueError):
            traffic = None

    rmse = None
    if not args.no_predict:
        sse = test.stats[:1].clone().to(red_dev if dist is not None else "cuda")
        if dist is not None:
            dist.all_reduce(sse)                       # every rank holds the squared error of its share of the test ratings
        rmse = float(np.sqrt(float(sse.item()) / n_test_total))

    if rank == 0:
        out = {
            "metric": "Gibbs sweeps/sec (both entities) + test RMSE, MovieLens-1M D=32",
            "value": round(replicas * args.steps / elapsed, 3),
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": source,
            "config": {"workload": f"BPMF MovieLens-1M 6040x3952, 500209 training ratings (500000 held out), D={D}, alpha=1.5, "
                                   f"step = rows of both entities + hyperpriors{'' if args.no_predict else ' + test prediction update'}"
                                   + (f"; {replicas} such units: the ratings stacked over {replicas} disjoint user blocks, "
                                      f"value = {replicas} x sweeps/s" if replicas > 1 else ""),
                       "num_latent": D, "burnin": args.warmup, "psamples": args.steps, "units_per_sweep": replicas,
                       "parallelism": (f"rows of each entity dealt over {world} GPUs, RCCL all-gather of the sampled rows per "
                                       f"half-sweep, test ratings split by user block") if world > 1 else "1 GPU"},
            "test_rmse": None if rmse is None else round(rmse, 5),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "k_sample_rows", "launches_timed": n_launch,
                         "avg_launch_us": round(1e3 * k1_ms / n_launch, 2),
                         "algorithmic_bytes_per_launch": int(k1_bytes / n_launch)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rd, D, args.seed)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
