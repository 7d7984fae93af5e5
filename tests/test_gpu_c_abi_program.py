"""examples/bpmf_c_abi.c: the whole BPMF loop of src/macau.jl:80-203 from a plain C program over include/bdf.h (no Python, no
torch in that process) -- the same chain, to the last bit, as the Python host over the same ABI."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayesiandatafusion.jl_amd", "csrc")


def _build(tmp_path):
    exe = str(tmp_path / "bpmf_c_abi")
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "bpmf_c_abi.c"), "-o", exe, "-L" + CSRC, "-lbdf_hip", "-lm",
                    "-Wl,-rpath," + CSRC], check=True)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_program_links_against_the_abi(tmp_path):
    """CPU: the example compiles with -Werror against include/bdf.h and links against the built library"""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("D", [32, 10])
def test_c_program_matches_python_host(tmp_path, D):
    import bdf_amd as B
    from bdf_amd import datasets
    from bdf_amd.engine import GibbsEngine
    n_rows, n_cols, nnz, sweeps = 20000, 3000, 400000, 12
    exe = _build(tmp_path)
    env = dict(os.environ)
    env.pop("BDF_RESERVE_CUS", None)
    out = subprocess.run([exe, str(n_rows), str(n_cols), str(nnz), str(D), str(sweeps)], check=True, capture_output=True,
                         text=True, env=env, timeout=300).stdout
    got = json.loads(out.strip().splitlines()[-1])
    rd = datasets.c4_relation_data(B, n_rows, n_cols, nnz)
    rel = rd.relations[0]
    eng = GibbsEngine(rd, D, seed=42)
    assert eng.native
    test = eng.test_pairs()
    burn = sweeps // 2
    for i in range(1, sweeps + 1):
        stats = eng.step(i, 0 if i <= burn else (1 if i == burn + 1 else 2), [1.0, 5.0], rel.class_cut)
    eng.sync()
    s = stats.cpu().numpy()
    assert got["test"] == test.n and got["train"] == rel.data.nnz()
    assert got["rmse"] == pytest.approx(float(np.sqrt(s[0] / test.n)), rel=0, abs=5e-7)     # printed with 6 decimals
    u0 = eng.ent[0].host("sample")[:, 0]
    assert got["row0_norm"] == pytest.approx(float(np.linalg.norm(u0)), rel=1e-11)
    assert 0.5 < got["rmse"] < 1.2
    eng.close()
