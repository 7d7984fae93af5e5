"""How does the CPU oracle's row phase scale with threads on this box (GPU box)?  MovieLens D=32 users' rows."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd import datasets
from oracle import oracle as O
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
r = rd.relations[0]
N = list(r.data.dims); D = 32
S = [np.zeros((N[0], D)), np.random.default_rng(0).standard_normal((N[1], D)) * 0.3]
t = O.Term(r.data.ids, r.data.values, N, 0, 1.5, r.data.valueMean(), [None, S[1]])
for nt in (1, 8, 16, 32, 64, 128, 256, 64, 16):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        O.sample_rows(D, N[0], [t], np.zeros(D), 5 * np.eye(D), 1, 2, 1, out=S[0], nthreads=nt)
        best = min(best, time.perf_counter() - t0)
    print(f"threads {nt:4d}: users' rows {best * 1e3:8.2f} ms")
