"""K1c against K1 as the GATHERED factor grows: 20,000 rows of ~100 observations (D = 32), the opposite entity of M rows
(2 MB ... 512 MB of factor); per-launch wall clock of entity 0's row launch alone, back to back (GPU box).
Run once per setting of BDF_K1_COL (128: K1c, 0: K1)."""
import os, sys, time
os.environ.setdefault("BDF_RESERVE_CUS", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bdf_amd as B
from bdf_amd._lib import check, lib
D = int(os.environ.get("D", "32"))
N1, per = int(os.environ.get("ROWS", "20000")), int(os.environ.get("PER_ROW", "100"))
MS = [int(x) for x in os.environ["MS"].split(",")] if os.environ.get("MS") else (8_000, 32_000, 128_000, 512_000, 2_000_000)
SORTED = os.environ.get("SORTED", "0") == "1"
for M in MS:
    rng = np.random.default_rng(M)
    rows = np.repeat(np.arange(N1, dtype=np.int64), per)
    cols = rng.integers(0, M, size=rows.size)
    if SORTED:
        cols = np.sort(cols.reshape(N1, per), axis=1).reshape(-1)
    ids = np.stack([rows + 1, cols + 1], axis=1)
    vals = rng.standard_normal(rows.size)
    rel = B.Relation((ids, vals), "r", [B.Entity("a"), B.Entity("b")], dims=[N1, M])
    rd = B.RelationData(rel)
    eng = B.GibbsEngine(rd, D, seed=1, device=0)
    for i in range(1, 4): eng.sweep(i)
    eng.sync()
    best = 1e9
    for rep in range(3):
        n = 20
        t0 = time.perf_counter()
        for i in range(n): check(lib().bdf_gibbs_rows_only(eng.gibbs, 0, 100 + i))
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    dsp = eng.rows_dispatch(0)
    print(f"D={D} rows={N1} per_row={per} sorted={int(SORTED)} BDF_K1_COL={os.environ.get('BDF_K1_COL')} M={M:8d} factor {M * D * 8 / 2**20:7.1f} MiB: {best * 1e6:8.1f} us per launch  dispatch {dsp}", flush=True)
    eng.close()
