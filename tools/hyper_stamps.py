"""Phase stamps of k_hyper_sample (diagnostic build: tools/ab_k1.sh build hst "-DBDF_HYPER_STAMPS")."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bdf_amd as B
from bdf_amd._lib import check, lib
ctx = B.Context(seed=1)
D, N = 32, 6040
rng = np.random.default_rng(0)
S = ctx.tensor(rng.standard_normal((N, D)))
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
sumU, UUt = ctx.zeros(D), ctx.zeros(D, D)
check(lib().bdf_hyper_sums(ctx.handle, D, N, p(S), None, p(sumU), p(UUt)))
mu0, Tinv, mu, Lam = ctx.zeros(D), ctx.tensor(np.eye(D)), ctx.zeros(D), ctx.zeros(D, D)
par = ctx.zeros(D + D * D + 16)
pack = ctx.zeros(lib().bdf_prior_pack_doubles(D))
draws = ctx.zeros(D * D + D)
for use_draws in (False, True):
    for rep in range(3):
        ctx.set_sweep(3 + rep)
        if use_draws:
            check(lib().bdf_hyper_draws(ctx.handle, D, N, float(D), 9, p(draws)))
        check(lib().bdf_hyper_sample(ctx.handle, D, N, p(sumU), p(UUt), p(mu0), 2.0, p(Tinv), float(D), 9, p(mu), p(Lam), p(par), p(pack),
                                     p(draws) if use_draws else None))
        ctx.sync()
    st = par.cpu().numpy()[D + D * D:].view(np.uint64).astype(np.int64)
    print("draws ahead" if use_draws else "draws inside", "phase ticks [muN, assemble, factor1, Zsolve, ZZt, factor2+mean, pack]:", np.diff(st[:7]), "total", st[6] - st[0])
import torch
for n in (6040, 3952):
    S2 = ctx.tensor(rng.standard_normal((n, D)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        e0.record(ctx.stream)
        for k in range(20):
            check(lib().bdf_hyper_sums(ctx.handle, D, n, p(S2), None, p(sumU), p(UUt)))
        e1.record(ctx.stream)
        ctx.sync()
    print(f"bdf_hyper_sums alone, N={n}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (back to back)")
