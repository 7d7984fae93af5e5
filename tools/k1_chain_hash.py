"""sha1 of both factor matrices after N iterations of BPMF on MovieLens (D = 32): two builds / launch modes of the row kernel
that claim bit-identical results print the same line (GPU box).   python tools/k1_chain_hash.py [N]"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd import datasets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
D = int(os.environ.get("D", "32"))
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
eng = B.GibbsEngine(rd, D, seed=1, device=0)
if os.environ.get("ITEM"):
    eng.ctx.set_item_size(int(os.environ["ITEM"]))
if os.environ.get("PIECE"):
    eng.ctx.set_piece_size(int(os.environ["PIECE"]))
for i in range(1, N + 1):
    eng.sweep(i)
eng.sync()
h = hashlib.sha1()
for e in rd.entities:
    h.update(np.ascontiguousarray(e.model.sample).tobytes())
print(f"chain hash after {N} iterations, D={D}: {h.hexdigest()}  (BDF_K1_QUEUE={os.environ.get('BDF_K1_QUEUE', '-')}, ITEM={os.environ.get('ITEM', '-')}, PIECE={os.environ.get('PIECE', '-')})")
eng.close()
