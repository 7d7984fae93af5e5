"""Quality figures of BASELINE.md section 4 on the GPU path: BPMF D=32 100+100, and Macau D=32 100+100 with the bundled
user / movie features."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B
from bdf_amd import datasets
for feats in (False, True):
    rd, src = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5, with_features=feats)
    t0 = time.time()
    res = B.macau(rd, burnin=100, psamples=100, num_latent=32, verbose=False, clamp=[1.0, 5.0], seed=11)
    print(f"{'Macau (29/18 binary features)' if feats else 'BPMF'} D=32 100+100: RMSE {res['RMSE']:.4f} accuracy {res['accuracy']:.4f} "
          f"ROC {res['ROC']:.4f}  ({time.time() - t0:.1f}s)")
