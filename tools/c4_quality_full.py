"""Configuration C4 at FULL size (10M x 1M, 100M observations, D = 64): held-out RMSE along a 30 + 30 run (GPU box).
Writes gpurun_out/r04_c4_quality.json."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd import datasets
from bdf_amd.engine import GibbsEngine
nb = int(os.environ.get("BURNIN", "30"))
rd = datasets.c4_relation_data(B)
rel = rd.relations[0]
tv = np.asarray(rel.test_vec.values)
eng = GibbsEngine(rd, 64, seed=5)
test = eng.test_pairs()
curve = []
t0 = time.time()
for i in range(1, 2 * nb + 1):
    stats = eng.step(i, 0 if i <= nb else (1 if i == nb + 1 else 2), [1.0, 5.0], rel.class_cut)
    if i in (1, 2, 5, 10, 20, 30, 31, 35, 40, 50, 60) or i == 2 * nb:
        eng.sync()
        s = stats.cpu().numpy()
        curve.append({"sweep": i, "rmse_running_mean": round(float(np.sqrt(s[0] / test.n)), 4), "rmse_this_sample": round(float(np.sqrt(s[1] / test.n)), 4)})
        print(curve[-1], flush=True)
eng.sync()
rec = {"workload": "configuration C4 at full size: synthetic 10,000,000 x 1,000,000, 100,000,000 observations (1% held out), BPMF D=64, alpha=2, "
                   f"{nb} burn-in + {nb} collected sweeps, seed 5; the low-rank sampler on (default)",
       "held_out": int(test.n), "value_std": round(float(tv.std()), 4),
       "mean_predictor_rmse": round(float(np.sqrt(np.mean((tv - rel.model.mean_value) ** 2))), 4), "noise_floor": 0.5774,
       "seconds": round(time.time() - t0, 1), "curve": curve}
os.makedirs(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "r04_c4_quality.json"), "w"), indent=1)
print(json.dumps(rec))
eng.close()
