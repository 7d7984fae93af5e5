"""macau() -- the Gibbs driver of the reference (src/macau.jl:3-255) over the device engine.

Same keyword surface and result keys as the reference.  Worker-process arguments (latent_pids, latent_blas_threads,
cg_pids) are accepted for drop-in compatibility; the GPU replaces the worker pool, so they only decide the
"latent_multi_threading" flag the reference reports (macau.jl:44-66, 253).  Extra keywords: seed, device.
"""
import math
import struct
import time

import numpy as np

from . import features as feat
from ._lib import ArgumentError
from .engine import GibbsEngine
from .relation_data import hasFeatures, numTest, toStr


def AUC_ROC(Ytrue, scores):
    """src/ROC.jl:1-11"""
    Ytrue = np.asarray(Ytrue, dtype=bool)
    perm = np.argsort(scores, kind="stable")
    roc_y = Ytrue[perm]
    if roc_y.sum() == 0 or (~roc_y).sum() == 0:
        return float("nan")
    stack_x = np.cumsum(roc_y) / roc_y.sum()
    stack_y = np.cumsum(~roc_y) / (~roc_y).sum()
    return float(np.sum((stack_x[1:] - stack_x[:-1]) * stack_y[1:]))


def makeClamped(x, clamp):
    """src/sampling.jl:99-106"""
    if len(clamp) == 0:
        return x
    return np.clip(x, clamp[0], clamp[1])


from .data_reading import read_binary_float32, write_binary_matrix  # noqa: E402,F401  (src/data_reading.jl:61-67, 93-99)


def macau(data, num_latent=10, lambda_beta=float("nan"), burnin=500, psamples=200, verbose=True, full_lambda_u=True,
          reset_model=True, compute_ff_size=6500, latent_pids=(1,), latent_blas_threads=1, cg_pids=(1,),
          full_prediction=False, rmse_train=False, tol=float("nan"), output="", output_beta=False, output_type="csv",
          clamp=(), f=False, seed=0, device=None, engine=None):
    if output_beta and not output:
        raise ArgumentError("To output samples of beta ('output_beta = true') you have to set also output prefix, "
                            "e.g., output = \"my_model\".")
    if output_type not in ("csv", "binary"):
        raise ArgumentError("output_type must be either \"csv\" or \"binary\".")
    clamp = [float(c) for c in clamp]

    verbose and print("Model setup")
    eng = engine
    if eng is None or reset_model:
        eng = GibbsEngine(data, num_latent, seed=seed, device=device, lambda_beta=lambda_beta,
                          compute_ff_size=compute_ff_size, full_lambda_u=full_lambda_u, tol=tol)
    data._engine = eng
    D = eng.D

    latent_multi_threading = (len(latent_pids) >= 1 and len(data.relations) == 1 and not hasFeatures(data.relations[0]))
    if verbose:
        if latent_multi_threading:
            print("Sampling of latent vectors: all rows of an entity in one GPU launch.")
        else:
            print("Sampling of latent vectors: general (multi-relation) GPU path.")

    rel = data.relations[0]
    haveTest = numTest(rel) > 0
    test = eng.test_pairs() if haveTest else None
    train = eng.train_pairs() if rmse_train else None
    f_output = []
    yhat_full = None
    if full_prediction:
        if hasFeatures(rel):
            raise ArgumentError("Prediction of all elements is not possible when Relation has features.")   # sampling.jl:92-94
        import torch
        yhat_full = torch.zeros(tuple(rel.data.dims), dtype=torch.float64, device=eng.ctx.device)
    rmse_avg = roc_avg = err_avg = float("nan")
    probe_avg = None

    verbose and print("Sampling")
    for i in range(1, burnin + psamples + 1):
        time0 = time.time()
        # relation models (alpha, relation beta) first, then rows, hyperpriors, beta (macau.jl:83-140), then the reporting
        # step on the test pairs (macau.jl:142-184); without side information all of it is one native call
        phase = 0 if i <= burnin else (1 if i == burnin + 1 else 2)
        stats = None
        if haveTest:
            stats = eng.step(i, phase, clamp, rel.class_cut)
        else:
            eng.sweep(i)
        facs = eng.factors_of(rel)
        if full_prediction and i > burnin:
            yhat_full += eng.pred_all(rel)                    # macau.jl:145-147: a plain dense product, on the device
        if i > burnin:
            if output:
                ndigits = int(math.floor(math.log10(psamples))) + 1
                nstr = str(i - burnin).rjust(ndigits, "0")
                for en in data.entities:
                    S = en.model.sample.astype(np.float32)
                    if output_type == "binary":
                        write_binary_matrix(f"{output}-{en.name}-{nstr}.binary", S)
                    else:
                        np.savetxt(f"{output}-{en.name}-{nstr}.csv", S, delimiter=",")
                    if output_beta and hasFeatures(en):
                        B = en.model.beta.astype(np.float32)
                        if output_type == "binary":
                            write_binary_matrix(f"{output}-{en.name}-{nstr}.beta.binary", B)
                        else:
                            np.savetxt(f"{output}-{en.name}-{nstr}.beta.csv", B, delimiter=",")
            if rmse_train:
                train.update(D, facs, rel.model.mean_value, phase, [], rel.class_cut)
            if i == burnin + 1 and verbose:
                print("--------- Burn-in complete, averaging posterior samples ----------")
            if callable(f):
                eng.sync()
                f_output.append(f(data))

        if verbose or i == burnin + psamples:
            eng.sync()
            eng.sync_host_scalars()
            if haveTest:
                s = stats.cpu().numpy()
                n = numTest(rel)
                rmse_avg = math.sqrt(s[0] / n)
                err_avg = s[2] / n
                probe_avg, _ = test.state()
                roc_avg = AUC_ROC(rel.test_label, -probe_avg)
            if verbose:
                estr = " ".join(toStr(en) for en in data.entities)
                rstr = " ".join(toStr(r) for r in data.relations)
                print(f"{i:3d}: ROC={roc_avg:6.4f} RMSE={rmse_avg:6.4f} | {estr} | {rstr} [{time.time() - time0:1.1f}s]")

    eng.sync()
    eng.sync_host_scalars()
    result = {
        "num_latent": num_latent,
        "burnin": burnin,
        "psamples": psamples,
        "lambda_beta": data.entities[0].lambda_beta,
        "RMSE": rmse_avg,
        "accuracy": err_avg,
        "ROC": roc_avg,
    }
    if full_prediction:
        result["predictions_full"] = (yhat_full / psamples).cpu().numpy()        # macau.jl:228-230
    if rmse_train:
        tavg, _ = train.state()
        result["RMSE_train"] = float(np.sqrt(np.mean((rel.data.getValues() - makeClamped(tavg, clamp)) ** 2)))
    if haveTest:
        avg, sq = test.state()
        if psamples >= 3:
            tmp = (sq - avg ** 2 * psamples) / (psamples - 1)
            tmp[tmp < 0] = 0
            stdev = np.sqrt(tmp)
        else:
            stdev = np.full(len(avg), np.nan)
        result["predictions"] = rel.test_vec.to_frame(pred=makeClamped(avg, clamp), stdev=stdev)
        import pandas as pd
        tc = np.zeros((numTest(rel), len(rel.entities)), dtype=np.int64)
        for mode in range(len(rel.entities)):
            rp = rel.data._rowptr[mode]
            ids = rel.test_vec.ids[:, mode].astype(np.int64)
            tc[:, mode] = rp[ids] - rp[ids - 1]
        result["train_counts"] = pd.DataFrame(tc, columns=[f"x{k + 1}" for k in range(tc.shape[1])])
    if callable(f):
        result["f_output"] = f_output
    result["latent_multi_threading"] = latent_multi_threading
    return result


# ---- prediction helpers of the reference (src/sampling.jl:9-97) on host copies of the factors --------------------
def pred(r, probe_vec=None, F=None):
    """pred(r) on the training table / pred(r, probe_vec) (sampling.jl:9-18) through the device kernel"""
    eng_rel = r._dev
    if eng_rel is None:
        raise ArgumentError("relation has no device state: run macau() first")
    from .engine import DevicePairs
    ctx = eng_rel.ctx
    ids = r.data.ids if probe_vec is None else np.asarray(getattr(probe_vec, "ids", probe_vec))[:, :len(r.entities)]
    pairs = DevicePairs(ctx, ids, np.zeros(len(ids)))
    facs = [e.model._dev.sample for e in r.entities]
    out = pairs.predict(facs[0].shape[1], facs, r.model.mean_value).cpu().numpy()
    pairs.close()
    return out


def pred_all(r):
    """pred_all(r) (sampling.jl:91-97): every cell of the relation, on host copies (test utility, not hot path)"""
    if hasFeatures(r):
        raise ArgumentError("Prediction of all elements is not possible when Relation has features.")
    S = [e.model.sample for e in r.entities]        # D x N_k
    if len(S) == 2:
        return S[0].T @ S[1] + r.model.mean_value
    letters = "abcdefg"[:len(S)]
    expr = ",".join(f"z{c}" for c in letters) + "->" + letters
    return np.einsum(expr, *S) + r.model.mean_value
