import faulthandler, os, sys
faulthandler.enable()
os.environ["BDF_NO_NATIVE"] = "1"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch, bdf_amd as B
from oracle import oracle as O
from test_gpu_engine_predict import _smoke_like
rd, rel = _smoke_like(B, 300, 200, 16, 6000, 500, seed=316)
eng = B.GibbsEngine(rd, 16, seed=42)
tp = eng.test_pairs()
for it in range(1, 6):
    print("sweep", it, flush=True)
    eng.sweep(it)
    print(" swept", flush=True)
    with torch.cuda.stream(eng.ctx.stream):
        pred = tp.predict(16, eng.factors_of(rel), rel.model.mean_value)
        got = pred.cpu().numpy()
    print(" predicted", flush=True)
    S = [eng.ent[j].host("sample").T for j in (0, 1)]
    exp = O.predict(rel.test_vec.ids, S, rel.model.mean_value)
    print(" oracle", np.abs(got - exp).max(), flush=True)
eng.close()
print("done")
