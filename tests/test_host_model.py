"""Host-side data model (IndexedDF / FastIDF / Entity / Relation / RelationData) against the reference's own test literals,
the C-ABI library's symbol table, and the sharded all-gather logic on CPU (gloo, world size 2).  No GPU needed: the only
calls into libbdf_hip.so are host-side (bdf_index_build, bdf_last_error).
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(B):
    header = open(os.path.join(ROOT, "include", "bdf.h")).read()
    declared = set(re.findall(r"\b(bdf_[a-z0-9_A-Z]+)\s*\(", header))
    declared -= {"bdf_ctx", "bdf_rel", "bdf_pairs", "bdf_feat", "bdf_term"}
    assert len(declared) >= 35
    lib = B.lib()                                   # resolves every symbol of the ctypes table or raises
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/bdf.h but not exported by libbdf_hip.so"
    assert set(B.declared_symbols()) == declared, set(B.declared_symbols()) ^ declared
    out = subprocess.run(["nm", "-D", "--defined-only", B.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (bdf_[a-zA-Z0-9_]+)", out))
    assert declared <= exported
    assert lib.bdf_version() >= 100


def test_no_gpu_fails_loudly(B):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ctypes as C
    h = C.c_void_p()
    rc = B.lib().bdf_ctx_create(0, None, C.c_uint64(1), C.byref(h))
    assert rc == -5 and b"no CPU path" in B.lib().bdf_last_error()
    with pytest.raises(B.NoGpuError):
        B.Context()


def test_indexeddf_reference_literals(B):
    """test/basic.jl:7-19"""
    X = B.IndexedDF({"A": [2, 2, 3], "B": [1, 3, 4], "C": [0.0, -1.0, 0.5]}, [4, 4])
    assert B.nnz(X) == 3
    assert B.getData(X, 1, 1)[0].shape == (0, 2) and B.getData(X, 1, 4)[0].shape == (0, 2)
    ids, vals = B.getData(X, 1, 2)
    assert ids.shape == (2, 2) and ids[:, 0].tolist() == [2, 2] and ids[:, 1].tolist() == [1, 3] and vals.tolist() == [0.0, -1.0]
    assert B.getData(X, 2, 2)[0].shape == (0, 2)
    assert [i.tolist() for i in X.index[0]] == [[], [1, 2], [3], []]
    assert [i.tolist() for i in X.index[1]] == [[1], [], [2], [3]]
    assert B.rep_int([2, 4, 1, 10], [3, 2, 1, 0]).tolist() == [2, 2, 2, 4, 4, 1]


def test_fastidf_and_types(B):
    """test/basic.jl:23-33"""
    X = B.IndexedDF({"A": [2, 2, 3], "B": [1, 3, 4], "C": [0.0, -1.0, 0.5]}, [4, 4])
    Xf = B.FastIDF(X)
    assert Xf.nnz() == 3
    assert Xf.getData(1, 1)[0].shape == (0, 2) and Xf.getData(1, 1)[1].shape == (0,)
    i, v = Xf.getData(1, 2)
    assert i[:, 0].tolist() == [2, 2] and i[:, 1].tolist() == [1, 3] and v.tolist() == [0.0, -1.0]
    X32 = B.FastIDF(B.IndexedDF((np.array([[1, 1], [2, 2], [3, 3]], dtype=np.int32), np.array([0.5, -0.1, 0.0], dtype=np.float32))))
    assert X32.Ti == np.int32 and X32.Tv == np.float32


def test_indexeddf_dims_and_remove(B):
    """test/basic.jl:36-49"""
    X2 = B.IndexedDF({"A": [2, 2, 3], "B": [1, 3, 4], "C": [0.0, -1.0, 0.5]}, (4, 4))
    assert X2.size() == (4, 4)
    X2a = B.IndexedDF({"A": [2, 2, 3], "B": [1, 1, 4], "C": [0.4, -1, -9]})
    assert X2a.size() == (3, 4)
    X3 = B.removeSamples(X2, [2])
    assert B.nnz(X3) == 2 and X3.size() == (4, 4)
    ids, _ = B.getData(X3, 1, 2)
    assert ids.shape == (1, 2) and B.getCount(X3, 1, 2) == 1
    with pytest.raises(B.BoundsError):
        B.IndexedDF({"A": [5], "B": [1], "C": [0.1]}, [4, 4])


def test_index_matches_oracle_bit_for_bit(B, O):
    rng = np.random.default_rng(0)
    for dims, nnz, dt in (([50, 40], 2000, np.int64), ([9, 7, 5], 300, np.int32), ([3, 3], 0, np.int64)):
        ids = np.stack([rng.integers(1, d + 1, nnz) for d in dims], axis=1).astype(dt).reshape(nnz, len(dims))
        idf = B.IndexedDF((ids, rng.standard_normal(nnz)), dims)
        ref = O.index_build(ids.astype(np.int64), dims) if nnz else None
        for m in range(len(dims)):
            if nnz:
                assert np.array_equal(idf._rowptr[m], ref[m][0]) and np.array_equal(idf._rowids[m], ref[m][1])
            else:
                assert idf._rowptr[m].tolist() == [0] * (dims[m] + 1)


def test_relation_entity_api(B):
    """test/basic.jl:52-101"""
    import scipy.sparse as sp
    a = {"A": [1, 2, 2, 3, 2], "B": [1, 3, 1, 4, 4], "v": [0.4, 1.0, -1.9, 1.4, 0.85]}
    r = B.Relation(a, "a")
    assert r.size() == (3, 4)
    r.F = np.array([[1.0, 2.5], [-1, -2], [0, 1], [3, -3]])
    B.assignToTest(r, [1, 4])
    assert np.array_equal(r.F, [[-1.0, -2], [0, 1]]) and np.array_equal(r.test_F, [[1.0, 2.5], [3, -3]])
    assert B.numTest(r) == 2 and B.numData(r) == 3
    e1, e2, e3 = B.Entity("e1"), B.Entity("e2"), B.Entity("e3")
    r2 = B.Relation(a, "r2", [e1, e2])
    assert e1.count == r2.size(1) and e2.count == r2.size(2) and r2.size() == (3, 4)
    B.setTest(r2, {"A": [1, 2], "B": [3, 4], "v": [0.1, -0.2]})
    assert r2.test_vec.shape == (2, 3) and B.numData(r2) == 5
    B.setTest(r2, sp.csc_matrix(([-0.4, 0.6, 0.7], ([2, 1, 1], [0, 1, 0])), shape=(3, 4)))
    assert B.numTest(r2) == 3 and B.numData(r2) == 5
    B.setPrecision(r2, 1.75)
    assert r2.model.alpha == 1.75
    r3 = B.Relation({"B": [1], "C": [5], "v": [0.1]}, "r3", [e2, e3])
    assert e3.count == 5 and r3.size() == (4, 5)
    with pytest.raises(B.ArgumentError):
        B.Relation({"A": [5], "C": [4], "v": [0.1]}, "r4", [e1, e3])


def test_relationdata_constructors(B):
    """test/basic.jl:97-101, test/custom_rd.jl:33-42, test/tensor.jl:16-23, docs multi-relation (note N1)"""
    import scipy.sparse as sp
    Y = sp.random(15, 10, 0.3, random_state=1, format="csc")
    rd = B.RelationData(Y, class_cut=0.5)
    assert [e.count for e in rd.entities] == [15, 10] and rd.relations[0].model.alpha == 5.0
    # findnz order of a SparseMatrixCSC is column-major (RelationData.jl:299-305)
    ids = rd.relations[0].data.ids
    assert np.all(np.diff(ids[:, 1]) >= 0)
    B.assignToTest(rd.relations[0], 2, rng=np.random.default_rng(0))
    assert B.numTest(rd.relations[0]) == 2 and len(rd.relations[0].test_label) == 2
    r2 = B.Relation(sp.random(100, 50, 0.01, random_state=2), "HPO2", [B.Entity("genes2"), B.Entity("pheno2")])
    assert r2.size() == (100, 50) and len(r2.entities) == 2
    rd2 = B.RelationData(r2)
    assert len(rd2.relations) == 1 and len(rd2.entities) == 2
    with pytest.raises(B.ArgumentError):
        B.RelationData(Y, feat1=np.zeros((14, 3)))
    # an entity shared by two relations is registered on both (the evident intent of addRelation!, note N1)
    a, b, c = B.Entity("a"), B.Entity("b"), B.Entity("c")
    rab = B.Relation({"a": [1, 2], "b": [1, 3], "v": [0.1, 0.2]}, "ab", [a, b])
    rac = B.Relation({"a": [2, 1], "c": [2, 2], "v": [0.3, 0.4]}, "ac", [a, c])
    rd3 = B.RelationData()
    B.addRelation(rd3, rab)
    B.addRelation(rd3, rac)
    assert len(a.relations) == 2 and len(rd3.entities) == 3 and len(rd3.relations) == 2
    bad = B.Relation({"a": [1], "b": [1], "v": [0.1]}, "bad", [a, b])
    bad.entities = [a]
    with pytest.raises(B.ArgumentError):
        B.addRelation(rd3, bad)


def test_sparse_bin_matrix_subsetting(B):
    """test/sbm.jl:6-32"""
    rows = np.concatenate([np.arange(1, 4), np.arange(2, 5), np.arange(1, 5)])
    cols = np.array([1, 1, 1, 2, 2, 2, 3, 3, 3, 3])
    m = B.SparseBinMatrix(rows, cols)
    assert m.size() == (4, 3)
    m2 = m[np.array([True, False, True, False]), :]
    assert m2.size() == (2, 3) and len(m2.rows) == 5
    assert m2.rows.tolist() == [1, 2, 2, 1, 2] and m2.cols.tolist() == [1, 1, 2, 3, 3]
    rng = np.random.default_rng(1)
    A = (rng.random((100, 50)) < 0.2).astype(float)
    I, J = np.nonzero(A)
    sbm = B.SparseBinMatrix(100, 50, I + 1, J + 1)
    z = np.zeros(100, dtype=bool)
    z[:20] = True; z[39] = True; z[59:80] = True
    assert np.array_equal(sbm[z, :].toarray(), A[z, :])
    with pytest.raises(B.DimensionMismatch):
        B.SparseBinMatrix(np.append(rows, 1), cols)
    csr = B.SparseBinMatrixCSR(rows, cols)
    assert csr.row_ptr.tolist() == [1, 3, 6, 9, 11] and csr.col_ind.tolist() == [1, 3, 1, 2, 3, 1, 2, 3, 2, 3]


def test_split_and_dataset_helpers(B):
    from bdf_amd import datasets
    ids = datasets.split_test_ids(1000, 400, seed=1)
    assert len(ids) == 400 and len(set(ids.tolist())) == 400 and ids.min() >= 1 and ids.max() <= 1000
    assert np.array_equal(ids, datasets.split_test_ids(1000, 400, seed=1))
    assert not np.array_equal(ids, datasets.split_test_ids(1000, 400, seed=2))
    # splitmix64 known values (reference implementation by Vigna: first outputs for state 0 and 1)
    assert int(datasets.splitmix64(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF
    if os.path.exists(datasets.MOVIELENS_PATH):
        d = datasets.load_movielens()
        assert d["X"].shape == (6040, 3952) and d["X"].nnz == 1000209
        assert d["Fu"].shape == (6040, 29) and d["Fv"].shape == (3952, 18)


_GLOO_WORKER = '''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
root, rank, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root)
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port
dist.init_process_group("gloo", rank=rank, world_size=world)
import importlib.util
import bdf_amd as B                                   # host-only pieces: no GPU in this test
from bdf_amd import _lib
import ctypes as C
rng = np.random.default_rng(7)
N, D = 1003, 6
degree = rng.integers(0, 200, N).astype(np.int64)
for chunks in (1, 3):
    pos = np.zeros(N, dtype=np.int32)
    cmax = C.c_int64(0)
    _lib.check(_lib.lib().bdf_layout_build(N, degree.ctypes.data_as(_lib.c_i64p), world, chunks, pos.ctypes.data_as(_lib.c_i32p), C.byref(cmax)))
    cmax = cmax.value
    nint = chunks * world * cmax
    assert cmax == -(-(-(-N // world)) // chunks) and len(set(pos.tolist())) == N and pos.max() < nint
    owner = (pos // cmax) % world
    chunk = (pos // cmax) // world
    # balance: rows dealt in falling order of degree, round-robin over the ranks, then over a rank's chunks
    load = [int(degree[owner == p].sum()) for p in range(world)]
    assert max(load) - min(load) <= degree.max()
    cl = [int(degree[(owner == rank) & (chunk == c)].sum()) for c in range(chunks)]
    assert max(cl) - min(cl) <= degree.max() * 2
    # the exchange: every rank "samples" its own rows (value = f(original id)), then one in-place all-gather per chunk
    expect = torch.zeros(nint, D, dtype=torch.float64)
    vals = torch.arange(N, dtype=torch.float64)[:, None] * 10.0 + torch.arange(D, dtype=torch.float64)[None, :] + 1.0
    expect[torch.as_tensor(pos.astype(np.int64))] = vals
    for rep in range(2):
        sample = torch.zeros(nint, D, dtype=torch.float64)
        mine = np.nonzero(owner == rank)[0]
        sample[torch.as_tensor(pos[mine].astype(np.int64))] = (rep + 1) * vals[torch.as_tensor(mine)]
        for c in range(chunks):
            region = sample[c * world * cmax:(c + 1) * world * cmax]
            dist.all_gather_into_tensor(region, region[rank * cmax:(rank + 1) * cmax].clone())       # bdf_allgather_rows: in place
        assert torch.equal(sample, (rep + 1) * expect), (rank, chunks, rep)
dist.destroy_process_group()
print("ok", rank)
'''


@pytest.mark.parametrize("world", [2, 4])
def test_layout_and_inplace_allgather_gloo(tmp_path, world):
    """the N > 1 path on CPU: bdf_layout_build (rows dealt over the ranks in falling order of degree, a rank's rows over its
    chunks; every chunk one contiguous rank-major region) and the exchange as an in-place all-gather per chunk --
    torch.distributed gloo here, ncclAllGather through bdf_allgather_rows on the GPUs"""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    port = str(29500 + (os.getpid() + 7 * world) % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in o, o


def test_sharded_relation_needs_no_gpu_to_be_checked(B):
    """bdf_layout_build on a relation's degrees: ranks and chunks partition the rows; positions are a bijection onto the
    non-padding rows"""
    import ctypes as C
    from bdf_amd import _lib
    deg = np.array([5, 0, 9, 9, 1, 3, 3, 2], dtype=np.int64)
    pos = np.zeros(8, dtype=np.int32)
    cmax = C.c_int64(0)
    _lib.check(_lib.lib().bdf_layout_build(8, deg.ctypes.data_as(_lib.c_i64p), 2, 2, pos.ctypes.data_as(_lib.c_i32p), C.byref(cmax)))
    assert cmax.value == 2
    # sorted by falling degree (stable): rows 2,3,0,5,6,7,4,1 -> rank s%2, index s//2 -> chunk (s//2)%2, slot (s//2)//2
    order = [2, 3, 0, 5, 6, 7, 4, 1]
    for s_, row in enumerate(order):
        p, r = s_ % 2, s_ // 2
        c, i = r % 2, r // 2
        assert pos[row] == (c * 2 + p) * 2 + i


def test_data_reading_formats(tmp_path):
    """src/data_reading.jl: byte layouts (Int64 headers, column-major payloads) and round trips; test/basic.jl:126-161 and
    test/beta_saving.jl:21-35 read posterior dumps back through read_binary_float32"""
    import struct
    import scipy.sparse as sp
    import bdf_amd as B
    rng = np.random.default_rng(0)
    X = rng.standard_normal((3, 5)).astype(np.float32)
    f = str(tmp_path / "m.binary")
    B.write_binary_matrix(f, X)
    raw = open(f, "rb").read()
    assert struct.unpack("<qq", raw[:16]) == (3, 5)
    assert np.array_equal(np.frombuffer(raw[16:], dtype="<f4"), X.T.ravel())          # column-major payload
    assert np.array_equal(B.read_binary_float32(f), X)
    Xi = rng.integers(-5, 5, (4, 2)).astype(np.int32)
    B.write_binary_int32(f, Xi)
    assert np.array_equal(B.read_binary_int32(f), Xi)
    S = sp.random(7, 6, density=0.4, random_state=1, format="csc")
    B.write_sparse_float64(f, S)
    assert (B.read_sparse_float64(f) != S).nnz == 0
    B.write_sparse_float32(f, S)
    r, c, v = B.read_sparse_float32(f)
    S32 = sp.csc_matrix((v, (r - 1, c - 1)), shape=S.shape)
    np.testing.assert_allclose(S32.toarray(), S.toarray(), rtol=1e-6)
    assert struct.unpack("<q", open(f, "rb").read(8))[0] == S.nnz
    B.write_sparse_binary_matrix(f, S)
    Sb = B.read_sparse_binary_matrix(f)
    assert np.array_equal(Sb.toarray() != 0, S.toarray() != 0) and Sb.shape == S.shape
    # text formats
    g = str(tmp_path / "rc.csv")
    open(g, "w").write("1,2\n3,1\n3,1\n")
    r, c = B.read_rowcol(g)
    assert r.tolist() == [1, 3, 3] and c.tolist() == [2, 1, 1] and r.dtype == np.int32
    assert B.read_sparse(g).toarray().tolist() == [[0, 1], [0, 0], [2, 0]]
    open(g, "w").write("mol1,77,5\nmol2,5,900\n")
    rows, cols, fp = B.read_ecfp(g)
    assert rows.tolist() == [1, 1, 2, 2] and cols.tolist() == [1, 2, 2, 3] and fp == {77: 1, 5: 2, 900: 3}
    assert B.filter_rare(sp.csc_matrix(np.array([[1, 0, 1], [1, 0, 0]])), 1).shape == (2, 2)
    m = str(tmp_path / "a.mtx")
    B.write_matrix_market(m, np.array([[1, 2, 0.5], [3, 1, -2.0]]))
    A = B.read_matrix_market(m)
    assert A.shape == (3, 2) and A[0, 1] == 0.5 and A[2, 0] == -2.0


def test_replicated_users_workload():
    """bench.py --gpus N (weak scaling): the rating matrix stacked over N disjoint user blocks; every block holds out the
    same entries and keeps the same training ratings"""
    import scipy.sparse as sp
    import bdf_amd as B
    from bdf_amd import datasets
    rng = np.random.default_rng(0)
    X = sp.random(30, 12, density=0.3, random_state=1, format="csc")
    X.data[:] = rng.integers(1, 6, X.nnz)
    tid = datasets.split_test_ids(X.nnz, 20, 1)
    base = B.Relation(X, "r", [B.Entity("u"), B.Entity("m")], class_cut=2.5)
    B.assignToTest(base, tid)
    big, ids = datasets.replicate_users(X, tid, 3)
    rel = B.Relation(big, "r", [B.Entity("u"), B.Entity("m")], class_cut=2.5)
    B.assignToTest(rel, ids)
    assert B.numTest(rel) == 3 * B.numTest(base) and B.numData(rel) == 3 * B.numData(base)

    def cells(vec):
        t = np.asarray(vec.ids).reshape(-1, 2)
        return sorted(zip(t[:, 0].tolist(), t[:, 1].tolist(), np.asarray(vec.values).tolist()))
    bt, btr = cells(base.test_vec), cells(base.data)
    assert cells(rel.test_vec) == sorted((u + 30 * k, m, v) for (u, m, v) in bt for k in range(3))
    assert cells(rel.data) == sorted((u + 30 * k, m, v) for (u, m, v) in btr for k in range(3))


def test_hot_kernels_use_no_scratch_and_keep_their_occupancy():
    """The build leaves every kernel's resource usage in csrc/*.o.res.  A helper that stops being inlined turns the row
    kernel's register arrays into scratch memory (seen once: the D=64 kernel went from 16 ms to 275 ms per sweep with every
    test still passing), so the hot kernels are pinned here: no scratch, no spills, and at least the occupancy DESIGN.md
    quotes."""
    import glob
    import re
    res = {}
    for f in glob.glob(os.path.join(ROOT, "bayesiandatafusion.jl_amd", "csrc", "*.o.res")):
        name = None
        for line in open(f):
            m = re.search(r"remark: \s*(Function Name|VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]): (\S+)", line)
            if not m:
                continue
            if m.group(1) == "Function Name":
                name = m.group(2)
                res[name] = {}
            elif name is not None:
                res[name][m.group(1).split(" ")[0]] = int(m.group(2))
    assert res, "no csrc/*.o.res: build with __graft_entry__.build() (make)"
    hot = {k: v for k, v in res.items() if re.search(r"6k_rowsILi|10k_rows_colILi|k_hyper_sampleILi|k_hyper_chainILi|k_hyper_partialILi|9k_predictILi|k_spmm_rm|k_dense_", k)}
    assert len(hot) >= 20, sorted(res)
    for k, v in hot.items():
        assert v["ScratchSize"] == 0 and v["VGPRs"] <= 256 and v.get("VGPRs Spill".split(" ")[0], 0) >= 0, (k, v)
    occ = {k: v["Occupancy"] for k, v in hot.items()}
    k = "_ZN12_GLOBAL__N_16k_rowsILi%dELb0ELb%dELb%dEEEv10SampleArgsNS_7PlanDevE"      # <DP, dump = false, two-mode, coded values>
    assert occ["_ZN12_GLOBAL__N_110k_rows_colILi32ELb1EEEv10SampleArgs10ColPlanDevj"] >= 2       # K1c at D = 32: the bench's kernel
    assert occ[k % (32, 1, 1)] >= 7       # one two-mode relation with coded values (ratings), D <= 32
    assert occ[k % (32, 1, 0)] >= 6       # two-mode variant, D <= 32
    assert occ[k % (32, 0, 0)] >= 5
    assert occ[k % (64, 1, 0)] >= 2 and occ[k % (64, 0, 0)] >= 2
    assert occ[k % (16, 1, 0)] >= 8


def test_synth_ratings_is_counter_based(B):
    """bdf_synth_ratings (configuration C4's generator, host-only): observation k does not depend on the range generated,
    ids stay in range, the column law is Zipf-like and about test_fraction of the observations are held out"""
    from bdf_amd import datasets
    r, c, v, h = datasets.synth_ratings(50_000, 4_000, 300_000, seed=777)
    r2, c2, v2, h2 = datasets.synth_ratings(50_000, 4_000, 1_000, seed=777, k_begin=123_456)
    s = slice(123_456, 124_456)
    assert np.array_equal(r[s], r2) and np.array_equal(c[s], c2) and np.array_equal(v[s], v2) and np.array_equal(h[s], h2)
    assert r.min() >= 1 and r.max() <= 50_000 and c.min() >= 1 and c.max() <= 4_000
    assert set(np.unique(v)) <= {1.0, 2.0, 3.0, 4.0, 5.0} and 3.2 < v.mean() < 3.8
    assert abs(h.mean() - 0.01) < 0.002
    cnt = np.bincount(c - 1, minlength=4_000).astype(float)
    # p(c) ~ 1 / (c + 100): the first 100 columns hold ln(2) / ln(41) of the mass
    assert abs(cnt[:100].sum() / cnt.sum() - np.log(2.0) / np.log(41.0)) < 0.01
    r3, _, _, _ = datasets.synth_ratings(50_000, 4_000, 1_000, seed=778)
    assert not np.array_equal(r3, r[:1000])
    rd = datasets.c4_relation_data(B, 5_000, 400, 40_000)
    rel = rd.relations[0]
    assert B.numData(rel) + B.numTest(rel) == 40_000 and rel.data.ids.dtype == np.int32 and rel.model.alpha == 2.0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as typed (WORLD_SIZE unset) starts two ranks through torch's launcher before anything
    touches the GPU runtime and relays rank 0's line (--rendezvous-only: the ranks meet over gloo and stop there)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BDF_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d == {"rendezvous": 2, "rank_sum": 1.0}


def test_feature_rows_subset_keeps_kind_and_content(B):
    """engine._take_rows (a rank's block of a relation's feature rows; the test rows a rank predicts): the chosen rows, in the
    chosen order, as a matrix of the same kind -- dense, scipy sparse, SparseMatrixCSR, SparseBinMatrix"""
    import scipy.sparse as sp
    from bdf_amd.engine import _take_rows
    from bdf_amd import features as feat
    rng = np.random.default_rng(4)
    A = rng.standard_normal((9, 5)) * (rng.random((9, 5)) < 0.5)
    rows0 = np.array([7, 2, 3])
    assert np.array_equal(_take_rows(A, rows0), A[rows0])
    assert np.array_equal(_take_rows(sp.csr_matrix(A), rows0).toarray(), A[rows0])
    r, c = np.nonzero(A)
    csr = feat.SparseMatrixCSR(r + 1, c + 1, A[r, c], 9, 5)
    sub = _take_rows(csr, rows0)
    dense = np.zeros((3, 5))
    dense[sub.rows - 1, sub.cols - 1] = sub.vals
    assert (sub.m, sub.n) == (3, 5) and np.array_equal(dense, A[rows0])
    bm = feat.SparseBinMatrix(9, 5, r + 1, c + 1)
    subb = _take_rows(bm, np.arange(2, 6))
    denseb = np.zeros((4, 5))
    denseb[subb.rows - 1, subb.cols - 1] = 1.0
    assert (subb.m, subb.n) == (4, 5) and np.array_equal(denseb, (A[2:6] != 0).astype(float))


def test_julia_binding_covers_every_product_export():
    """julia/BDFHip.jl (the ccall layer a BayesianDataFusion.jl maintainer would add; no Julia in this image, so it is checked
    as text): every entry point include/bdf.h declares is bound, except the ones listed here as diagnostics / test hooks /
    host-language plumbing a Julia host has no use for; and the structs it mirrors have the C structs' fields in order."""
    import re
    h = open(os.path.join(ROOT, "include", "bdf.h")).read()
    jl = open(os.path.join(ROOT, "julia", "BDFHip.jl")).read()
    exports = sorted(set(re.findall(r"\b(bdf_[a-z0-9_]+)\s*\(", h)))
    not_needed = {
        "bdf_version", "bdf_ctx_stream", "bdf_ctx_advance_sweep",                                   # plumbing
        "bdf_comm_create_host", "bdf_comm_size",                                                     # the one-GPU test rig's transport
        "bdf_ctx_set_gather", "bdf_row_system", "bdf_normals", "bdf_philox", "bdf_rows_unfinished",  # parity hooks
        "bdf_event_create", "bdf_event_destroy", "bdf_event_elapsed_us", "bdf_ctx_time_next_rows", "bdf_ctx_time_next_hyper",
        "bdf_ctx_span_next_rows", "bdf_gibbs_span_rows", "bdf_comm_peer_stats", "bdf_gibbs_time_rows", "bdf_gibbs_rows_only", "bdf_gibbs_contexts", "bdf_gibbs_recorded", "bdf_gibbs_set_recorded",   # measurement
        "bdf_synth_ratings",                                                                         # the bench's generator
    }
    unbound = [n for n in exports if (":" + n) not in jl and n not in not_needed]
    assert not unbound, unbound
    assert not [n for n in not_needed if n not in exports], "stale names in the exemption list"

    def c_fields(name):
        end = h.index("} " + name + ";")
        body = h[h.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        body = re.sub(r"struct \{.*?\} (\w+)\[[^\]]*\];", r"X \1;", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                for v in decl.split(","):
                    out.append(re.sub(r"\[.*", "", v.strip().split()[-1].lstrip("*")))
        return out

    def jl_fields(name):
        body = re.search(r"struct " + name + r"\b(.*?)\nend", jl, re.S).group(1)
        body = re.sub(r"#.*", "", body)
        return [f.split("::")[0].strip() for f in re.split(r"[;\n]", body) if "::" in f]

    assert jl_fields("Term") == c_fields("bdf_term")
    assert jl_fields("GibbsRelation") == c_fields("bdf_gibbs_relation")
    assert jl_fields("GibbsEntity") == c_fields("bdf_gibbs_entity")


def test_no_dpp_read_inside_a_hazard_window():
    """the row kernels' DPP instructions are inline assembly (the compiler's hazard recogniser does not look inside) and runs of
    them drop the `s_nop` on the strength of "the source was written long before": the build leaves the device assembly of
    those translation units in csrc/*.s and tools/dpp_hazard_check.py walks it -- no vector (2 wait states) or matrix (18)
    instruction may write a DPP source inside its window.  ~18,000 DPP instructions."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dpp_hazard_check", os.path.join(ROOT, "tools", "dpp_hazard_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "bayesiandatafusion.jl_amd", "csrc", "*.s")))
    assert {os.path.basename(f) for f in files} >= {"k_sample_rows.s", "k_rows_lr.s", "k_rows_col.s", "k_hyper.s", "k_block.s"}, "build with __graft_entry__.build() (make)"
    total = 0
    for f in files:
        bad, n = mod.check(f)
        assert not bad, bad[:5]
        total += n
    assert total > 10000
    # and the checker does see a hazard when there is one
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as t:
        t.write("f:\n\tv_add_f64 v[2:3], v[4:5], v[6:7]\n\tv_fmac_f64_dpp v[8:9], v[2:3], v[10:11] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    bad, n = mod.check(t.name)
    os.unlink(t.name)
    assert n == 1 and len(bad) == 1
