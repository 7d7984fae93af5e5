#!/usr/bin/env python3
"""bench.py -- Gibbs sweeps/sec (both entities) + test RMSE, BPMF on MovieLens-1M, D=32, fp64 (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one Gibbs iteration of src/macau.jl:80 on the device: latent rows of users, hyperprior of users, latent rows of
movies, hyperprior of movies, and the test-set prediction update of macau.jl:142-184 (inside the timed region; the
reporting-only metrics are not) -- one native call, bdf_gibbs_sweep.  W warm-up steps are the burn-in; the K timed steps
are the posterior samples, so the RMSE printed is that of the posterior-mean prediction after W + K iterations on the
500,000 held-out ratings.

The JSON line has two measurements:
 * value / ms_per_step: the BASELINE metric.  N = 1 is exactly the BASELINE configuration.  N > 1 is WEAK scaling of it:
   MovieLens is a ~120 us sweep, far too small to split, so the N-GPU workload is N MovieLens-sized units -- the rating
   matrix stacked over N disjoint user blocks that rate the same movies (datasets.replicate_users) -- value = N x sweeps/s.
 * "c4": STRONG scaling on BASELINE configuration 4, the synthetic 10M x 1M relation with 100M observations at D = 64
   (datasets.c4_relation_data / bdf_synth_ratings): the same relation on 1, 2, 4, 8 GPUs, c4.sweeps_per_s.  (--c4-nnz etc.
   shrink it; --no-c4 skips it.)
In both, with N > 1: one process per GPU; every rank holds the observations of its own rows and a replica of both factor
matrices (rows shared out by bdf_layout_build), samples its rows and takes part in an in-place RCCL all-gather per
half-sweep (bdf_allgather_rows); the test ratings are split over the ranks.

roofline: K1 (k_rows) algorithmic bytes per launch over its mean launch duration from HIP events attached to
the kernel dispatches (on the launch stream), from the timed region and -- so that at least 200 launches are averaged --
from further sweeps after it; `avg_launch_us_alone` / `frac_alone`: the same kernel with nothing beside it (back-to-back
launches on the row stream, wall clock).  cpu_baseline: the CPU oracle (a C port of the reference algorithm, OpenMP over rows like
the reference's latent_pids workers) timed on this box.
"""
import argparse
import hashlib
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
    r = rd.relations[0]
    nthreads = _host_threads()
    multi, rows_only, n = cpu_rows_sweeps(r.data.ids, r.data.values, list(r.data.dims), r.model.alpha, D, seed, None, nthreads, budget_s, r.data.valueMean())
    single, _, _ = cpu_rows_sweeps(r.data.ids, r.data.values, list(r.data.dims), r.model.alpha, D, seed, 2, 1, 0.0, r.data.valueMean())
    return {"value": round(1.0 / multi, 4), "unit": "sweeps/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} full Gibbs sweeps (rows of both entities + hyperpriors) of the same MovieLens D={D} training set, "
                      f"oracle/bdf_oracle.c with OpenMP over rows on {_threads_note(nthreads)} (the row phase alone: {1.0 / rows_only:.1f} sweeps/s); "
                      f"single thread: {1.0 / single:.4f} sweeps/s",
            "value_1thread": round(1.0 / single, 4), "value_rows_only": round(1.0 / rows_only, 4)}


def _cpu_quota():
    """CPUs' worth of time the container may use (cgroup cpu.max / cfs quota), or None"""
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                return int(q) / int(per)
        except (OSError, ValueError, IndexError):
            pass
    return None


def _host_threads():
    """threads for the CPU baseline: the hardware threads the process may run on, capped by the container's CPU quota (more
    threads than the quota only get throttled: 256 threads on a 16-CPU quota ran the row phase 15 x slower than 16 did)
    (not omp_get_max_threads: the OpenMP runtime may have been loaded while the process was pinned to one core)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = _cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q + 0.5)))
    return max(1, min(n, 1024))


def _threads_note(nthreads):
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = _cpu_quota()
    return f"{nthreads} threads" + (f" (the container's CPU quota, cpu.max = {q:g} CPUs, of the box's {n} hardware threads)" if q is not None and q < n else "")


def cpu_rows_sweeps(ids, vals, dims, alpha, D, seed, n_sweeps, nthreads, budget_s=0.0, mean=None):
    """(seconds per sweep, seconds of it in the row phase, sweeps timed) of the CPU oracle on a two-mode relation: rows of both
    entities + hyperprior draws; one untimed sweep first; n_sweeps None: as many as fit budget_s (3 .. 200)"""
    import numpy as np
    from oracle import oracle as O
    N = list(dims)
    idx = O.index_build(ids, N)
    S = [np.zeros((N[0], D)), np.zeros((N[1], D))]
    mu = [np.zeros(D), np.zeros(D)]
    Lam = [5.0 * np.eye(D), 5.0 * np.eye(D)]
    mean = float(np.mean(vals)) if mean is None else float(mean)
    # the relation's view of each entity, built once: the rows are sampled in place (a term never reads its own entity's rows)
    terms = [O.Term(ids, vals, N, j, alpha, mean, [None if k == j else S[k] for k in (0, 1)], index=idx) for j in (0, 1)]
    t_rows = [0.0]

    def sweep(it):
        for j in (0, 1):
            t0 = time.perf_counter()
            O.sample_rows(D, N[j], [terms[j]], mu[j], Lam[j], seed, it, j + 1, out=S[j], nthreads=nthreads)
            t_rows[0] += time.perf_counter() - t0
            mu_N, beta_N, T_N, nu_N = O.hyper_params(S[j], np.zeros(D), 2.0, np.eye(D), float(D))
            mu[j], Lam[j] = O.hyper_draw(mu_N, beta_N, T_N, nu_N, seed, it, j + 1)

    sweep(1)                      # untimed: first touch, and a non-zero factor state
    t_rows[0] = 0.0
    t0 = time.perf_counter()
    n = 0
    while (n < n_sweeps) if n_sweeps is not None else (n < 3 or (time.perf_counter() - t0 < budget_s and n < 200)):
        sweep(2 + n)
        n += 1
    return (time.perf_counter() - t0) / n, t_rows[0] / n, n


def c4_cpu_baseline(rel4, D, budget_s=10.0):
    """The CPU oracle beside configuration C4.  A full sweep of 11M rows at D = 64 takes the oracle minutes, so it is timed on
    two samples of the SAME relation with every sampled row's complete observation list -- the first users (uniform rows,
    ~10 observations each) and a block of mid-popularity items (hundreds to thousands each) -- and the sweep is extrapolated
    with the cost model t = a rows + b observations fitted to the two (stated as such in `sample`)."""
    import numpy as np
    from oracle import oracle as O
    ids, vals = np.asarray(rel4.data.ids), np.asarray(rel4.data.values, dtype=np.float64)
    Nu, Ni = [int(x) for x in rel4.data.dims]
    nnz = len(vals)
    nthreads = _host_threads()
    mean = float(vals.mean())
    mu, Lam = np.zeros(D), 5.0 * np.eye(D)

    def timed(mode, lo, hi):
        """rows lo < id <= hi of `mode` with their complete observation lists (the other mode's ids compacted)"""
        m = (ids[:, mode] > lo) & (ids[:, mode] <= hi)
        sub = np.empty((int(m.sum()), 2), dtype=np.int64, order="F")
        sub[:, mode] = ids[m, mode] - lo
        other, inv = np.unique(ids[m, 1 - mode], return_inverse=True)
        sub[:, 1 - mode] = inv + 1
        dims = [0, 0]
        dims[mode], dims[1 - mode] = hi - lo, len(other)
        fac = [None, None]
        fac[1 - mode] = np.full((len(other), D), 0.05)
        t = O.Term(sub, vals[m], dims, mode, 2.0, mean, fac)
        O.sample_rows(D, min(dims[mode], 256), [t], mu, Lam, 5, 1, mode + 1, row_end=min(dims[mode], 256), nthreads=nthreads)     # first touch
        t0 = time.time()
        O.sample_rows(D, dims[mode], [t], mu, Lam, 5, 2, mode + 1, nthreads=nthreads)
        return time.time() - t0, dims[mode], int(m.sum())

    # users: a probe sets the sample so that it takes about half the budget
    tp, rp, _ = timed(0, 0, min(Nu, 20_000))
    n1 = int(min(Nu, max(20_000, 0.5 * budget_s / max(tp, 1e-3) * rp)))
    t1, r1, o1 = timed(0, 0, n1)
    c0 = min(Ni // 2, 5000)
    tp, rp, _ = timed(1, c0, min(Ni, c0 + 500))
    n2 = int(min(Ni - c0, max(500, 0.5 * budget_s / max(tp, 1e-3) * rp)))
    t2, r2, o2 = timed(1, c0, c0 + n2)
    # t = a rows + b observations
    det = r1 * o2 - r2 * o1
    a = (t1 * o2 - t2 * o1) / det if det else t1 / max(r1, 1)
    b = (r1 * t2 - r2 * t1) / det if det else 0.0
    if a <= 0 or b <= 0:          # degenerate fit: rows only / observations only
        a, b = t1 / max(r1, 1), t2 / max(o2, 1)
    sweep_s = a * (Nu + Ni) + b * 2 * nnz
    return {"value": round(1.0 / sweep_s, 5), "unit": "sweeps/s", "cores": nthreads, "kind": "port", "ms_per_sweep": round(1e3 * sweep_s, 1),
            "sample": f"EXTRAPOLATED: oracle/bdf_oracle.c, OpenMP over rows on {_threads_note(nthreads)}, timed on {r1} users ({o1} observations, {t1:.2f} s) "
                      f"and {r2} items ({o2} observations, {t2:.2f} s) of this relation with their complete observation lists; cost model "
                      f"t = {a * 1e6:.2f} us x rows + {b * 1e9:.1f} ns x observations applied to {Nu + Ni} rows and 2 x {nnz} observations "
                      f"(row sampling only: the hyperprior draws are not in it)"}


def c3_block(B, datasets, D, device, sweeps=20):
    """Macau on MovieLens with the dense 6040 x 500 user features of configuration C3: one native call per iteration (uhat,
    rows, hyperpriors with the feature terms, beta); the direct solve and the forced conjugate gradients"""
    out = {"workload": f"Macau MovieLens-1M + dense user side information 6040 x 500 (iid N(0,1), seed 4242), D={D}, 5 + {sweeps} sweeps, "
                       f"no prediction update; ms per sweep"}
    # (cg_correlated: SURVEY M-C3's second feature matrix, Z W + 0.1 E with 20 dominant directions -- the conditioning CG feels)
    for ff_size, key, kind in ((6500, "ff", "iid"), (0, "cg", "iid"), (0, "cg_correlated", "correlated")):
        rd, _ = datasets.c3_relation_data(B, kind)
        eng = B.GibbsEngine(rd, D, seed=1, device=device, compute_ff_size=ff_size)
        eng.warm_device(30.0)
        for i in range(1, 6):
            eng.sweep(i)
        eng.sync()
        t0 = time.perf_counter()
        for i in range(6, 6 + sweeps):
            eng.sweep(i)
        eng.sync()
        dt = (time.perf_counter() - t0) / sweeps
        it = eng.ent[0].cg_iters.cpu().numpy()
        b3 = sum(eng.k1_algorithmic_bytes(j) for j in range(len(eng.ent))) + (2 * int(it.max()) + 3) * 6040 * 500 * 8
        out[key] = {"ms_per_sweep": round(1e3 * dt, 4), "sweeps_per_s": round(1.0 / dt, 1), "native_iteration": bool(eng.native),
                    "roofline": config_roofline("c3_" + ("cg" if key.startswith("cg") else key), b3, 1e3 * dt,
                                                flops_sweep=sum(eng.k1_algorithmic_flops(j) for j in range(len(eng.ent))) + (2 * int(it.max()) + 3) * 2.0 * 6040 * 500 * D),
                    "features": "iid N(0,1), seed 4242" if kind == "iid" else "correlated: Z W + 0.1 E, Z 6040 x 20 (seeds 4243 / 4244)",
                    "solver": "direct: F'F = Q diag(s) Q' once, beta = Q ((Q' rhs) ./ (s + lambda)) per iteration" if ff_size else "conjugate gradients (compute_ff_size=0)",
                    "cg_iterations_last_sweep": [int(it.min()), int(it.max())]}
        eng.close()
    return out


def mref_block(B, device, cpu):
    """The reference's own benchmark shape (test/benchmark_parallel_latent.jl:8-61): sprand(1_500_000, 1000, 0.01)-like
    relation (uniform positions, U(0,1) values, seed 1500), BPMF with D = 10 and 30, 2 + 2 iterations, 50 test entries"""
    import numpy as np
    rng = np.random.default_rng(1500)
    N, M = 1_500_000, 1000
    nnz = int(N * M * 0.01)
    key = np.unique(rng.integers(0, N * M, size=int(nnz * 1.01)))[:nnz]
    ids = np.stack([key // M + 1, key % M + 1], axis=1)
    vals = rng.random(len(key))
    out = {"workload": f"uniform {N} x {M}, {len(key)} observations U(0,1) (seed 1500), BPMF, 2 burn-in + 2 timed iterations, 50 test entries"}
    for D in (10, 30):
        rel = B.Relation((ids, vals), "r", [B.Entity("rows"), B.Entity("cols")], dims=[N, M])
        B.assignToTest(rel, np.arange(1, 51))
        rd = B.RelationData(rel)
        eng = B.GibbsEngine(rd, D, seed=1, device=device)
        test = eng.test_pairs()
        for i in (1, 2):
            eng.step(i, 0, [], rel.class_cut)
        eng.sync()
        t0 = time.perf_counter()
        for i in (3, 4):
            eng.step(i, 1 if i == 3 else 2, [], rel.class_cut)
        eng.sync()
        dt = (time.perf_counter() - t0) / 2
        bytes_sweep = sum(eng.k1_algorithmic_bytes(j) for j in (0, 1))
        blk = {"ms_per_sweep": round(1e3 * dt, 3), "sweeps_per_s": round(1.0 / dt, 2), "algorithmic_gb_per_sweep": round(bytes_sweep / 1e9, 2),
               "algorithmic_tb_per_s": round(bytes_sweep / dt / 1e12, 3),
               "roofline": config_roofline(f"mref_d{D}", bytes_sweep, 1e3 * dt, flops_sweep=sum(eng.k1_algorithmic_flops(j) for j in (0, 1))),
               # rows drawn by the low-rank sampler (same conditional distribution as the reference's map, other values: DESIGN 4)
               "lowrank_rows": [eng.lowrank_rows(j) for j in (0, 1)]}
        train_ids, train_vals = np.asarray(rel.data.ids), np.asarray(rel.data.values, dtype=np.float64)
        eng.close()
        if cpu:
            nt = _host_threads()
            s_all, s_rows, _ = cpu_rows_sweeps(train_ids, train_vals, [N, M], rel.model.alpha, D, 1, 2, nt)
            blk["cpu_baseline"] = {"value": round(1.0 / s_all, 4), "unit": "sweeps/s", "cores": nt, "kind": "port",
                                   "sample": f"2 full sweeps (rows of both entities + hyperpriors) of the same relation after one untimed, "
                                             f"oracle/bdf_oracle.c with OpenMP over rows on {_threads_note(nt)} (the row phase alone: {1.0 / s_rows:.3f} sweeps/s)"}
        out[f"d{D}"] = blk
    return out


def pin_to_quiet_core(share, shares):
    """The host thread that enqueues the sweeps runs ~15 HIP calls per 100 us sweep: 40 us when it stays on one quiet core, 60 us
    when the scheduler moves it about, 80-90 us when it ends up away from the memory the runtime's queues live in -- and then the
    sweep is host-bound (8.9k instead of 10k sweeps/s on the same GPU; tools/numa_probe.sh).  Pin the process to the block of
    four neighbouring cores that was idlest over the last 100 ms (SMT siblings counted) -- one for the enqueueing thread, the
    others for the runtime's own threads: pinned to ONE core they share it with the enqueueing thread and 4 runs in 10 are
    host-bound; with four, none of 12 was.  -> (first core, previous affinity) or (None, None); BDF_BENCH_PIN=0 turns it off,
    BDF_BENCH_PIN_CORES sets the width."""
    if os.environ.get("BDF_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None, None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        mine = [c for k, c in enumerate(allowed) if k % max(shares, 1) == share]
        if len(allowed) < 4 or not mine:
            return None, None

        def snap():
            out = {}
            for line in open("/proc/stat"):
                if line.startswith("cpu") and line[3].isdigit():
                    f = line.split()
                    v = [int(x) for x in f[1:9]]
                    out[int(f[0][3:])] = (sum(v), v[3] + v[4])          # total, idle + iowait
            return out

        a = snap(); time.sleep(0.1); b = snap()
        busy = {c: (b[c][0] - a[c][0]) - (b[c][1] - a[c][1]) for c in b if c in a}

        def sibling(c):
            try:
                sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip().replace("-", ",").split(",")
                return [int(x) for x in sib if int(x) != c]
            except Exception:
                return []

        load = lambda c: busy.get(c, 0) + sum(busy.get(x, 0) for x in sibling(c))
        width = int(os.environ.get("BDF_BENCH_PIN_CORES", "4"))
        if width <= 1:
            core = min(mine, key=lambda c: (load(c), c))
            os.sched_setaffinity(0, {core})
            return core, set(allowed)
        # a block of `width` neighbouring cores (one for the enqueueing thread, the others for the runtime's own threads)
        aset = set(allowed)
        starts = [c for c in mine if all((c + k) in aset for k in range(width))]
        if not starts:
            return None, None
        core = min(starts, key=lambda c: (sum(load(c + k) for k in range(width)), c))
        os.sched_setaffinity(0, {core + k for k in range(width)})
        return core, set(allowed)
    except Exception:
        return None, None


TRAFFIC_JSON = "profiles/r06_hbm_traffic.json"
PMC_JSON = "profiles/r06_pmc_k_rows.json"
CONFIG_KERNELS_JSON = "profiles/r06_config_kernels.json"
FP64_PEAK_TFLOPS = 78.6            # MI355X fp64 vector = matrix peak (the matrix and the vector instructions share the pipe)


def k1_source_sha1():
    """sha1 over the row kernel's source AND every local header it includes, transitively (k_sample_rows.hip, bdf_common.h,
    wave_linalg.h, c_layout_chol.h, include/bdf.h, ...): the identity of the code the PMC passes profiled"""
    import re
    csrc = os.path.join(ROOT, "bayesiandatafusion.jl_amd", "csrc")
    seen, todo = {}, [os.path.join(csrc, "k_sample_rows.hip"), os.path.join(csrc, "k_rows_col.hip")]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen:
            continue
        seen[f] = open(f, "rb").read()
        for inc in re.findall(rb'^\s*#\s*include\s+"([^"]+)"', seen[f], flags=re.M):
            for base in (os.path.dirname(f), csrc, os.path.join(ROOT, "include")):
                cand = os.path.join(base, inc.decode())
                if os.path.exists(cand):
                    todo.append(cand)
                    break
    h = hashlib.sha1()
    for f in sorted(seen):
        h.update(os.path.relpath(f, ROOT).encode() + b"\0" + seen[f])
    return h.hexdigest()


def recorded_traffic():
    """HBM bytes of one K1 launch from the committed PMC passes (tools/profile_round.sh: FETCH_SIZE and WRITE_SIZE in
    separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -- NOT measured by this run: returned
    only while the row kernel's source (with every header it includes) is the one the passes ran on, with the file named
    next to it."""
    try:
        d = json.load(open(os.path.join(ROOT, TRAFFIC_JSON)))
        if d.get("k1_source_sha1") != k1_source_sha1():
            return None, None
        return d["k1_traffic_bytes_per_launch"]["hbm_bytes_fetch_doubled"], TRAFFIC_JSON + " (rocprofv3 --pmc passes of this workload on this kernel source; not measured by this run)"
    except (OSError, KeyError, ValueError):
        return None, None


def recorded_pipe(launch_us):
    """FP64-pipe occupancy of the headline row kernel from the committed counter passes (tools/profile_round_r05.sh; NOT measured by
    this run, returned only for the kernel source they ran on): SQ_ACTIVE_INST_VALU (quad-cycles summed over the SIMDs) and
    SQ_VALU_MFMA_BUSY_CYCLES per launch, over 1,024 SIMDs x the launch's cycles at 2.4 GHz."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_JSON)))
        if d.get("k1_source_sha1") != k1_source_sha1():
            return None
        k = next(v for kk, v in d["kernels"].items() if "k_rows_col" in kk)
        valu = 4.0 * k["SQ_ACTIVE_INST_VALU"]["mean"]
        mfma = k.get("SQ_VALU_MFMA_BUSY_CYCLES", {"mean": 0.0})["mean"]
        if not launch_us or launch_us <= 0:           # (--k1-min-launches 0: no launch was timed)
            return None
        cyc = launch_us * 1e-6 * 2.4e9 * 1024
        return {"valu_busy_cycles_per_launch": round(valu), "mfma_busy_cycles_per_launch": round(mfma),
                "valu_instructions_per_launch": round(k["SQ_INSTS_VALU"]["mean"]) if "SQ_INSTS_VALU" in k else None,
                "frac": round((valu + mfma) / cyc, 3), "of": "1,024 SIMDs x launch duration x 2.4 GHz", "source": PMC_JSON + " (not measured by this run)"}
    except (OSError, KeyError, ValueError, StopIteration):
        return None


def config_roofline(name, bytes_sweep, ms_per_sweep, extra=None, flops_sweep=None):
    """`roofline` of one of the other configurations: algorithmic bytes (and flops) of the row launches per sweep over
    the measured sweep.  The roof that binds is the one SURVEY 8d derives from the arithmetic intensity against the ridge
    78.6 TF / 8 TB/s = 9.8 flop/B: HBM below it (C2, C3, M-ref: 4.4-6 flop/B), the fp64 pipe above (C4: 17.7); both figures are
    given.  The dominant kernel and its rocprofv3 average come from the committed profile of that configuration
    (CONFIG_KERNELS_JSON, written by tools/profile_round_r06.sh: not measured by this run)."""
    gbs = bytes_sweep / 1e9 / (ms_per_sweep / 1e3)
    r = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_sweep": int(bytes_sweep), "traffic": None}
    if flops_sweep:
        tf = flops_sweep / 1e12 / (ms_per_sweep / 1e3)
        ai = flops_sweep / max(bytes_sweep, 1)
        r.update({"algorithmic_flops_per_sweep": int(flops_sweep), "flop_per_byte": round(ai, 2),
                  "hbm_gbs": round(gbs, 2), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                  "fp64_tflops": round(tf, 2), "fp64_frac": round(tf / FP64_PEAK_TFLOPS, 4)})
        if ai > FP64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS:          # above the ridge: the fp64 pipe is the roof
            r.update({"bound": "fp64", "achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_PEAK_TFLOPS, 4),
                      "note": "flops by the reference's map (sampling.jl:200-212: D^2 n + D^3/3 per row); the low-rank sampler does fewer for short rows"})
    try:
        r["dominant_kernel"] = dict(json.load(open(os.path.join(ROOT, CONFIG_KERNELS_JSON)))[name], source=CONFIG_KERNELS_JSON + " (not measured by this run)")
    except (OSError, KeyError, ValueError):
        r["dominant_kernel"] = None
    if extra:
        r.update(extra)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--num-latent", type=int, default=32)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--k1-event-every", type=int, default=16,
                    help="time the K1 launches of every n-th step (HIP events attached to the kernel dispatch)")
    ap.add_argument("--k1-min-launches", type=int, default=200, help="K1 launches averaged for the roofline (further sweeps after the timed region)")
    ap.add_argument("--replicas", type=int, default=0, help="user blocks of the workload (default: one per GPU)")
    ap.add_argument("--device-warmup-ms", type=float, default=60.0,
                    help="engine set-up before the first warm-up step: full iterations whose results are discarded (the chain's state is "
                         "put back bit for bit), for this long; brings the device to its working state; 0 = none")
    ap.add_argument("--no-c4", action="store_true", help="skip the strong-scaling measurement on configuration C4")
    ap.add_argument("--no-c4-uniform", action="store_true", help="skip configuration C4's uniform-column variant (SURVEY M-C4)")
    ap.add_argument("--no-c3", action="store_true", help="skip the C3 block (Macau with dense side information; one GPU only)")
    ap.add_argument("--no-mref", action="store_true", help="skip the block on the reference's own benchmark shape (one GPU only)")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 block (3-mode tensor + matrix sharing an entity with binary sparse "
                                                          "features; strong scaling over the ranks)")
    ap.add_argument("--c5-sizes", default="", help="C5 (tests): nA,nB,nC,nT,n1,n2,n_feat,feat_per_row instead of the configuration's sizes")
    ap.add_argument("--c5-sweeps", type=int, default=10, help="C5: burn-in sweeps, then as many timed sweeps with the prediction update")
    ap.add_argument("--c4-rows", type=int, default=10_000_000)
    ap.add_argument("--c4-cols", type=int, default=1_000_000)
    ap.add_argument("--c4-nnz", type=int, default=100_000_000)
    ap.add_argument("--c4-latent", type=int, default=64)
    ap.add_argument("--c4-sweeps", type=int, default=5, help="warm-up and timed sweeps of the C4 measurement (5 + 5: SURVEY M-C4)")
    ap.add_argument("--dump-state", default="", help="(tests) PREFIX: every rank leaves every entity's sample (the reference's row order), mu and "
                                                     "Lambda of the main workload and of the C4 / C5 blocks in PREFIX.<block>.rank<r>.npz")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="test hook: every rank joins the process group, all-reduces its rank, rank 0 prints it; no GPU is touched")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: start the N ranks ourselves -- one process per GPU through torch's launcher, as
        # a CHILD process (nothing in this process has touched the GPU runtime yet, and nothing will) -- and relay its output
        # and exit code.  The form `python -m torch.distributed.run ... bench.py --gpus N` keeps working: it sets WORLD_SIZE.
        # (--standalone: the launcher's own rendezvous on a port IT binds -- no port picked here and released before use)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # before anything initialises the GPU runtime: its queues and signals are then allocated from the memory of the core the
    # enqueueing thread stays on (a thread pinned later, away from where the runtime was initialised, enqueues at 80 us per sweep)
    # (one rank only: with several, RCCL's proxy threads would inherit the one-core mask and fight the enqueueing thread for it)
    host_core, full_affinity = pin_to_quiet_core(0, 1) if world == 1 else (None, None)

    import numpy as np
    import torch

    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rendezvous_only:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous": world, "rank_sum": float(t.item())}), flush=True)
        dist.destroy_process_group()
        return
    import bdf_amd as B
    from bdf_amd import datasets
    # test rig: BDF_DIST_BACKEND=gloo runs all ranks on GPU 0 with the exchange staged through the host (RCCL needs one
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

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def my_share(n):
        """the test pairs this rank predicts: a contiguous slice"""
        return np.arange(n * rank // world, n * (rank + 1) // world)

    def dump_state(engine, block):
        if not args.dump_state:
            return
        st = {}
        for j, e in enumerate(engine.ent):
            st[f"S{j}"] = e.host("sample")          # D x N, the reference's row order whatever the layout
            st[f"mu{j}"] = e.host("mu")
            st[f"Lam{j}"] = e.host("Lambda")
            if getattr(e, "numF", 0):
                st[f"beta{j}"] = e.host("beta")
        np.savez(f"{args.dump_state}.{block}.rank{rank}.npz", **st)

    # ---- the BASELINE metric: MovieLens (N > 1: N stacked units) ------------------------------------------------------------
    D = args.num_latent
    replicas = args.replicas if args.replicas > 0 else world
    rd, source = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5, replicas=replicas)
    rel = rd.relations[0]
    # continuity with the rounds before the set-up warm-up existed: the same --warmup + --steps on a throw-away
    # engine WITHOUT it, first thing in the process -- a cold device, as those rounds' driver lines were measured
    value_cold = None
    if world == 1 and args.device_warmup_ms > 0 and not os.environ.get("BDF_BENCH_NO_COLD"):
        eng0 = B.GibbsEngine(rd, D, seed=args.seed, device=local_rank)
        eng0.test_pairs()
        eng0.register_test([1.0, 5.0], rel.class_cut)
        for i in range(1, args.warmup + 1):
            eng0.step(i, 0, [1.0, 5.0], rel.class_cut)
        eng0.sync()
        t0c = time.perf_counter()
        for k in range(args.steps):
            eng0.step(args.warmup + 1 + k, 1 if k == 0 else 2, [1.0, 5.0], rel.class_cut)
        eng0.sync()
        value_cold = args.steps / (time.perf_counter() - t0c)
        eng0.close()
        del eng0
        from bdf_amd.relation_data import EntityModel
        for en in rd.entities:              # (the model the throw-away engine attached to the entities: the real one makes its own)
            en.model = EntityModel()

    eng = B.GibbsEngine(rd, D, seed=args.seed, device=local_rank, shard=(rank, world))
    n_test_total = len(np.asarray(rel.test_vec.values))
    test = eng.test_pairs(subset=my_share(n_test_total) if world > 1 else None)
    clamp = [1.0, 5.0]

    # set-up, not a step: the device is brought to its working state by full iterations whose results are discarded -- the
    # chain's state is put back bit for bit, nothing of it advances (engine.warm_device; tools/region_idle_probe.py: a
    # 20-iteration region right after idle time or after row launches alone runs 5-10 % slower than in sustained work)
    import gc
    gc.collect()          # (as timeit does: no collector pause inside the 2 ms timed region -- a collection here, before the
    gc.disable()          #  device warm-up, not between the warm-up steps and the region, where it would be idle time)
    eng.register_test(clamp, rel.class_cut)
    eng.warm_device(args.device_warmup_ms)
    for i in range(1, args.warmup + 1):
        eng.step(i, 0, clamp, rel.class_cut)
    eng.sync()
    fence()
    # the timed region: exactly --steps iterations, nothing else (the K1 launch timers are attached to sweeps AFTER it)
    eng.k1_events = None
    stamps = [] if os.environ.get("BDF_BENCH_DEBUG") else None
    t0 = time.perf_counter()
    for k in range(args.steps):
        eng.step(args.warmup + 1 + k, 1 if k == 0 else 2, clamp, rel.class_cut)
        if stamps is not None:
            stamps.append(time.perf_counter())
    t_enq = time.perf_counter()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    gc.enable()
    eng.sync()
    if stamps is not None:
        per = [round(1e6 * (b - a), 1) for a, b in zip([t0] + stamps[:-1], stamps)]
        print(f"[bench] timed region: enqueue per step (us) {per}; enqueue total {1e6 * (t_enq - t0):.0f} us, "
              f"final wait {1e6 * (t0 + elapsed - t_enq):.0f} us", file=sys.stderr)
    rmse = None
    sse = test.stats[:1].clone().to(red_dev if dist is not None else "cuda")
    if dist is not None:
        dist.all_reduce(sse)                       # every rank holds the squared error of its share of the test ratings
    rmse = float(np.sqrt(float(sse.item()) / n_test_total))
    dump_state(eng, "main")

    # K1 roofline: the launches timed inside the timed region, then further sweeps (every launch timed) up to the minimum
    # (i) device-side spans: first wave's start to last wave's end of every row launch (s_memrealtime, K1c only), no event packets
    it = args.warmup + args.steps
    span_us = []
    if eng.native and args.k1_min_launches > 0:
        eng.k1_span_begin(args.k1_min_launches)
        for _ in range((args.k1_min_launches + len(eng.ent) - 1) // len(eng.ent)):
            it += 1
            eng.sweep(it, 2)
        span_us = eng.k1_span_result()
    # (ii) HIP events on the kernels' dispatch packets (every K1 variant; the events themselves lengthen the launch)
    eng.k1_events = []
    eng.k1_event_every = 1
    t_ev = time.perf_counter()
    it_ev0 = it
    while len(eng.k1_events) < args.k1_min_launches and it < it_ev0 + 2000:
        it += 1
        eng.sweep(it, 2)
    eng.sync()
    k1_ms = sum(t.elapsed_us() for (_, t) in eng.k1_events) / 1e3
    k1_bytes = sum(eng.k1_algorithmic_bytes(j) for (j, _) in eng.k1_events) / max(world, 1)
    k1_flops = sum(eng.k1_algorithmic_flops(j) for (j, _) in eng.k1_events) / max(world, 1)
    n_launch = max(len(eng.k1_events), 1)
    launch_us_events = 1e3 * k1_ms / n_launch
    launch_us = (sum(u for _, u in span_us) / len(span_us)) if span_us else launch_us_events
    pipe = recorded_pipe(launch_us)
    achieved = (k1_bytes / n_launch / 1e9) / (launch_us / 1e6) if launch_us > 0 else 0.0
    eng.k1_events = None
    traffic, traffic_source = recorded_traffic() if (world == 1 and replicas == 1 and D == 32) else (None, None)
    # the same kernel with nothing beside it: back-to-back launches alternating the entities on the row stream, wall clock (one
    # GPU only; in the sweep a launch also holds the wait for the prior inside the kernel and shares the CUs with the prediction
    # update).  After the timed region: the chain's state is no longer used.
    alone_us = None
    if world == 1 and args.k1_min_launches > 0:
        eng.sync()
        torch.cuda.synchronize()
        n_alone = 200
        ta = time.perf_counter()
        from bdf_amd._lib import check as _check, lib as _lib
        for i in range(n_alone):
            if eng.native:          # the launch the sweep makes (same kernel variant), alone
                _check(_lib().bdf_gibbs_rows_only(eng.gibbs, i % len(eng.ent), 1_000_000 + i))
            else:
                eng.ctx.set_sweep(1_000_000 + i)
                eng.sample_entity(i % len(eng.ent))
        enq_us = 1e6 * (time.perf_counter() - ta) / n_alone
        eng.sync()
        alone_us = 1e6 * (time.perf_counter() - ta) / n_alone
        if os.environ.get("BDF_BENCH_DEBUG"):
            print(f"[bench] K1 alone: enqueue {enq_us:.1f} us per launch, total {alone_us:.1f} us", file=sys.stderr)

    if full_affinity:
        os.sched_setaffinity(0, full_affinity)            # the CPU baseline and the C4 generator use every core
    out = None
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
                                   f"step = rows of both entities + hyperpriors + test prediction update (one native call); "
                                   f"BEFORE the --warmup steps the engine's set-up runs {args.device_warmup_ms:g} ms of full "
                                   f"iterations whose results are discarded (device_warmup_ms; the value without it: "
                                   f"value_without_device_warmup)"
                                   + (f"; {replicas} such units: the ratings stacked over {replicas} disjoint user blocks, "
                                      f"value = {replicas} x sweeps/s" if replicas > 1 else ""),
                       "num_latent": D, "burnin": args.warmup, "psamples": args.steps, "units_per_sweep": replicas,
                       "host_core": host_core,
                       "device_warmup_ms": args.device_warmup_ms if eng.gibbs else 0.0,
                       "device_warmup": "set-up: full iterations for that long, results discarded, the chain's state put back bit for bit",
                       "parallelism": (f"rows of each entity shared out over {world} GPUs (a rank holds its rows' observations only), "
                                       f"in-place all-gather of the sampled rows per half-sweep, test ratings split over the ranks; "
                                       f"transport: {eng.comm.transport if eng.comm is not None else 'none'}")
                       if world > 1 else "1 GPU"},
            "test_rmse": None if rmse is None else round(rmse, 5),
            "value_without_device_warmup": None if value_cold is None else round(value_cold, 3),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "k_rows_col (k_rows_col.hip: K1c) when D in 17..32, else k_rows (k_sample_rows.hip)", "launches_timed": n_launch,
                         # avg_launch_us: first wave's start to last wave's end of the launch, stamped BY THE KERNEL (s_memrealtime, one
                         # atomic min / max per wave: bdf_gibbs_span_rows) on sweeps after the timed region -- what rocprofv3 reports as
                         # the kernel's duration, with no event packets around the dispatch; by_events: HIP events riding on the dispatch
                         # packets (what rounds 1-4 quoted: the events lengthen the launch by a few us)
                         "avg_launch_us": round(launch_us, 2), "avg_launch_us_timer": "in-kernel s_memrealtime span" if span_us else "HIP events",
                         "launches_spanned": len(span_us), "avg_launch_us_by_events": round(launch_us_events, 2),
                         "fp64_pipe": pipe,
                         # the same as scalars (what a parser that keeps only numbers and strings sees): the FP64 pipe's busy share of the
                         # launch from the committed counters, the launch's flops by the reference's map against the 78.6 TF peak, and the
                         # counters' HBM traffic over the algorithmic bytes (<< 1: the gathered factor is L2-resident, nothing re-read)
                         "fp64_valu_busy_frac": None if pipe is None else pipe["frac"],
                         "fp64_flops_frac": (round(k1_flops / n_launch / (launch_us * 1e-6) / (FP64_PEAK_TFLOPS * 1e12), 4)
                                             if n_launch and launch_us > 0 else None),
                         "traffic_over_algorithmic": None if traffic is None or not k1_bytes else round(traffic / (k1_bytes / n_launch), 3),
                         "counters_schedule": "the committed PMC passes ran with BDF_RESERVE_CUS=0 BDF_NO_POLL=1 (rocprofv3 serialises the streams; CU-masked "
                                              "streams crash its teardown): all 1,024 SIMDs, event hand-overs -- the launch time they are divided by is this run's",
                         "algorithmic_bytes_per_launch": int(k1_bytes / n_launch),
                         "avg_launch_us_alone": None if alone_us is None else round(alone_us, 2),
                         "frac_alone": None if alone_us is None else round((k1_bytes / n_launch / 1e9) / (alone_us / 1e6) / HBM_PEAK_GBS, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rd, D, args.seed)
    eng.close()
    del eng, rd, rel, test

    # ---- strong scaling on configuration C4 ---------------------------------------------------------------------------------
    def c4_variant(zipf_offset, columns, with_cpu):
        c4 = {"workload": f"synthetic {args.c4_rows} x {args.c4_cols}, {args.c4_nnz} observations (1% held out), BPMF D={args.c4_latent}, "
                          f"alpha=2, {args.c4_sweeps}+{args.c4_sweeps} sweeps (SURVEY M-C4: bdf_synth_ratings seed 777, {columns} columns)",
              "n_gpus": world, "scaling": "strong"}
        try:
            t0 = time.time()
            rd4 = datasets.c4_relation_data(B, args.c4_rows, args.c4_cols, args.c4_nnz, zipf_offset=zipf_offset)
            t_gen = time.time() - t0
            rel4 = rd4.relations[0]
            t0 = time.time()
            eng4 = B.GibbsEngine(rd4, args.c4_latent, seed=5, device=local_rank, shard=(rank, world))
            n4 = len(np.asarray(rel4.test_vec.values))
            test4 = eng4.test_pairs(subset=my_share(n4) if world > 1 else None)
            eng4.step(1, 0, clamp, rel4.class_cut)
            eng4.sync()
            t_setup = time.time() - t0
            for i in range(2, args.c4_sweeps + 1):
                eng4.step(i, 0, clamp, rel4.class_cut)
            eng4.sync()
            fence()
            t0 = time.perf_counter()
            for k in range(args.c4_sweeps):
                eng4.step(args.c4_sweeps + 1 + k, 1 if k == 0 else 2, clamp, rel4.class_cut)
            fence()
            el4 = max_over_ranks(time.perf_counter() - t0)
            eng4.sync()
            sse4 = test4.stats[:1].clone().to(red_dev if dist is not None else "cuda")
            if dist is not None:
                dist.all_reduce(sse4)
            bytes_sweep = sum(eng4.k1_algorithmic_bytes(j) for j in range(2))
            rmse4 = float(np.sqrt(float(sse4.item()) / max(n4, 1)))
            mp4 = float(np.sqrt(np.mean((np.asarray(rel4.test_vec.values) - rel4.model.mean_value) ** 2)))
            c4.update({"sweeps_per_s": round(args.c4_sweeps / el4, 3), "ms_per_sweep": round(1e3 * el4 / args.c4_sweeps, 3),
                       "roofline": config_roofline("c4", bytes_sweep, 1e3 * el4 / args.c4_sweeps, flops_sweep=sum(eng4.k1_algorithmic_flops(j) for j in range(2))),
                       "test_rmse": round(rmse4, 5),
                       # (a chain this short has not left its start: above the mean predictor the figure says nothing about the model)
                       "test_rmse_above_mean_predictor": bool(rmse4 > mp4),
                       # what the held-out RMSE is to be read against: the spread of the held-out values = the RMSE of predicting
                       # their mean, and the generator's noise floor sqrt(0.5^2 + 1/12) (rounded N(., 0.5)); after only
                       # c4-sweeps + c4-sweeps sweeps the chain is still near the first (profiles/r04_c4_quality.json: the same
                       # relation run 30 + 30)
                       "value_std": round(float(np.asarray(rel4.test_vec.values).std()), 4),
                       "mean_predictor_rmse": round(float(np.sqrt(np.mean((np.asarray(rel4.test_vec.values) - rel4.model.mean_value) ** 2))), 4),
                       "noise_floor": 0.5774,
                       "algorithmic_gb_per_sweep": round(bytes_sweep / 1e9, 2),
                       "algorithmic_tb_per_s": round(bytes_sweep / (el4 / args.c4_sweeps) / 1e12, 3),
                       "lowrank_rows": [eng4.lowrank_rows(j) for j in range(2)],
                       "exchange": None if eng4.comm is None else dict(zip(("peer_exchanges", "peer_bytes_pulled_per_rank"), eng4.comm.peer_stats()),
                                                                      transport=eng4.comm.transport),
                       "chunks": eng4.layouts[0].chunks, "generate_s": round(t_gen, 1), "setup_s": round(t_setup, 1),
                       "device_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)})
            dump_state(eng4, "c4" if zipf_offset else "c4_uniform")
            eng4.close()
            if with_cpu and not args.no_cpu_baseline and world == 1:         # (the CPU figures: one process only, as the main line's)
                del eng4, test4
                c4["cpu_baseline"] = c4_cpu_baseline(rel4, args.c4_latent)
        except Exception as e:      # noqa: BLE001 -- the BASELINE metric above must still be reported
            c4["error"] = f"{type(e).__name__}: {e}"
            print(f"[bench] rank {rank}: the C4 block failed: {c4['error']}", file=sys.stderr, flush=True)
        return c4

    if not args.no_c4:
        c4 = c4_variant(100.0, "Zipf-like", True)
        if out is not None:
            out["c4"] = c4
        # SURVEY M-C4's other column law: uniform columns (every item ~100 observations), same sizes, no CPU leg
        if not args.no_c4_uniform:
            c4u = c4_variant(0.0, "uniform", False)
            if out is not None:
                out["c4_uniform"] = c4u
    # ---- configuration C5: tensor + matrix sharing an entity with binary sparse features (strong scaling) -------------------
    if not args.no_c5:
        z5 = dict(zip(("nA", "nB", "nC", "nT", "n1", "n2", "n_feat", "feat_per_row"),
                      [int(x) for x in (args.c5_sizes or "100000,64,1000,500,5000000,1000000,50000,50").split(",")]))
        c5 = {"workload": f"entity A ({z5['nA']} rows, binary sparse features {z5['nA']} x {z5['n_feat']}, {z5['feat_per_row']} per row) shared by a "
                          f"3-mode relation A x B x C ({z5['nB']} x {z5['nC']}; {z5['n1']} cells of a planted rank-8 model, noise 0.1, 1% held out) and a "
                          f"2-mode relation A x T ({z5['nT']}; {z5['n2']} cells); Macau D=32, alpha 5 / 2, beta by conjugate gradients; "
                          f"{args.c5_sweeps}+{args.c5_sweeps} sweeps (SURVEY M-C5)",
              "n_gpus": world, "scaling": "strong"}
        try:
            rd5, info5 = datasets.c5_relation_data(B, **z5)
            rel5 = rd5.relations[0]
            eng5 = B.GibbsEngine(rd5, 32, seed=3, device=local_rank, compute_ff_size=0, shard=(rank, world))
            n5 = len(np.asarray(rel5.test_vec.values))
            test5 = eng5.test_pairs(subset=my_share(n5) if world > 1 else None)
            for i in range(1, args.c5_sweeps + 1):
                eng5.step(i, 0, [], rel5.class_cut)
            eng5.sync()
            fence()
            t0 = time.perf_counter()
            for k in range(args.c5_sweeps):
                eng5.step(args.c5_sweeps + 1 + k, 1 if k == 0 else 2, [], rel5.class_cut)
            fence()
            el5 = max_over_ranks(time.perf_counter() - t0)
            eng5.sync()
            sse5 = test5.stats[:1].clone().to(red_dev if dist is not None else "cuda")
            if dist is not None:
                dist.all_reduce(sse5)
            it5 = eng5.ent[0].cg_iters.cpu().numpy()
            bytes5 = sum(eng5.k1_algorithmic_bytes(j) for j in range(len(eng5.ent)))
            # the beta update's sparse products: 2 per CG iteration (F p, F' t) + 3 (right-hand side, uhat, ...): gathered bytes =
            # nonzeros x 8 D per product; HBM-minimal = indices + the two dense operands once
            nnzF, nA5, nF5 = z5["nA"] * z5["feat_per_row"], z5["nA"], z5["n_feat"]
            n_prod = 2 * int(it5.max()) + 3
            gathered = n_prod * nnzF * 8 * 32
            minimal = n_prod * (nnzF * 4 + (nA5 + nF5) * 8 * 32)
            ms5 = 1e3 * el5 / args.c5_sweeps
            c5.update({"sweeps_per_s": round(args.c5_sweeps / el5, 2), "ms_per_sweep": round(ms5, 3),
                       "roofline": config_roofline("c5", bytes5 + minimal, ms5, {
                           "sparse_products_per_sweep": n_prod, "gathered_bytes_per_product": nnzF * 8 * 32,
                           "hbm_minimal_bytes_per_product": nnzF * 4 + (nA5 + nF5) * 8 * 32,
                           "note": "the sweep is its sparse products (k_spmm_rm16): gathered rows of a 12.8 / 25.6 MB dense operand come from the "
                                   "Infinity Cache (a 4 MiB XCD L2 holds a third / a sixth of it), whose gather rate (8.6 TB/s, MI355X_MICROARCH.md) "
                                   "is the bound the product runs at; `achieved` prices the HBM-minimal bytes"}),
                       "test_rmse": round(float(np.sqrt(float(sse5.item()) / max(n5, 1))), 5), "value_std": round(info5["value_std"], 4),
                       "noise": info5["noise"], "native_iteration": bool(eng5.native),
                       "cg_iterations_last_sweep": [int(it5.min()), int(it5.max())], "beta_columns_per_rank": -(-32 // world)})
            dump_state(eng5, "c5")
            eng5.close()
            del eng5, test5, rd5, rel5
        except Exception as e:      # noqa: BLE001
            c5["error"] = f"{type(e).__name__}: {e}"
            print(f"[bench] rank {rank}: the C5 block failed: {c5['error']}", file=sys.stderr, flush=True)
        if out is not None:
            out["c5"] = c5
    # ---- configuration C3 and the reference's own benchmark shape (one GPU) ------------------------------------------------
    if world == 1 and out is not None:
        for name, skip, fn in (("c3", args.no_c3, lambda: c3_block(B, datasets, D, local_rank)),
                               ("mref", args.no_mref, lambda: mref_block(B, local_rank, not args.no_cpu_baseline))):
            if skip:
                continue
            try:
                out[name] = fn()
            except Exception as e:      # noqa: BLE001
                out[name] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
