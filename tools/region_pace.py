"""Pace of short timed regions over the life of a process (GPU box): after W warm-up sweeps, R regions of S sweeps each,
every region bracketed by a full synchronisation like bench.py's; prints each region's us per sweep and the GPU clock
(pp_dpm_sclk) where readable.  Usage: region_pace.py [W] [R] [S] [idle_ms between regions]"""
import glob, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bdf_amd as B
from bdf_amd import datasets

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
S = int(sys.argv[3]) if len(sys.argv) > 3 else 20
idle_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
warm_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0


def sclk():
    out = []
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            out += [l.strip() for l in open(p) if "*" in l]
        except OSError:
            pass
    return ";".join(out)


rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
rel = rd.relations[0]
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
test = eng.test_pairs()
it = 0
t_first = time.perf_counter()
eng.warm_device(warm_ms)
print(f"device warm-up {warm_ms} ms took {1e3 * (time.perf_counter() - t_first):.1f} ms")
for i in range(W):
    it += 1
    eng.step(it, 0, [1.0, 5.0], rel.class_cut)
eng.sync(); torch.cuda.synchronize()
for r in range(R):
    if idle_ms:
        time.sleep(idle_ms / 1e3)
    t0 = time.perf_counter()
    for k in range(S):
        it += 1
        eng.step(it, 2 if r or k else 1, [1.0, 5.0], rel.class_cut)
    t1 = time.perf_counter()
    eng.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"region {r:3d} at {1e3 * (t0 - t_first):8.1f} ms: {1e6 * (t2 - t0) / S:7.1f} us/sweep (enqueue {1e6 * (t1 - t0) / S:6.1f}, final wait {1e6 * (t2 - t1):6.0f} us)  sclk {sclk()}")
eng.close()
