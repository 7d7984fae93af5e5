"""Entity / Relation / RelationData -- host mirror of src/RelationData.jl of the reference.

Only the data model lives here (what the user builds before calling macau); every numeric step of the Gibbs
sweep is done by libbdf_hip.so through engine.GibbsEngine.  Names follow the reference with the trailing `!`
dropped (assignToTest!, setTest!, setPrecision!, addRelation!, normalizeFeatures!, normalizeRows!).
Entity and mode numbers in accessors are 1-based as in the reference.
"""
import math

import numpy as np

from . import features as feat
from ._lib import ArgumentError
from .indexed_df import IndexedDF, _split_table


class EntityModel:
    """EntityModel (RelationData.jl:14-40).  Arrays live on the device once macau() has initialised the model;
    the attributes below return host copies in the reference's orientation (sample is D x N)."""

    def __init__(self):
        self._dev = None          # engine.EntityState

    def _get(self, name):
        if self._dev is None:
            raise AttributeError("model not initialised: call macau() (reset!) first")
        return self._dev.host(name)

    sample = property(lambda self: self._get("sample"))
    mu = property(lambda self: self._get("mu"))
    Lambda = property(lambda self: self._get("Lambda"))
    beta = property(lambda self: self._get("beta"))
    uhat = property(lambda self: self._get("uhat"))
    mu0 = property(lambda self: self._get("mu0"))
    WI = property(lambda self: self._get("WI"))
    b0 = property(lambda self: self._dev.b0)
    nu0 = property(lambda self: self._dev.nu0)


class Entity:
    """Entity(name; F=zeros(0,0), lambda_beta=1.0) (RelationData.jl:42-63)"""

    def __init__(self, name, F=None, lambda_beta=1.0):
        self.F = F
        self.FF = None
        self.use_FF = False
        self.relations = []
        self.count = 0
        self.name = str(name)
        self.modes = []
        self.modes_other = []
        self.lambda_beta = float(lambda_beta)
        self.lambda_beta_sample = True
        self.mu = 1.0          # hyper-prior for lambda_beta
        self.nu = 1e-3
        self.model = None

    def __repr__(self):
        s = f"[Entity] {self.name}: {self.count:6d} "
        if hasFeatures(self):
            lam = "sample" if self.lambda_beta_sample else f"{self.lambda_beta:1.1f}"
            return s + f"with {feat.feature_shape(self.F)[1]} features (λ = {lam})"
        return s + "with no features"


def hasFeatures(x):
    return not feat.isempty(x.F)


def toStr(x):
    if isinstance(x, Entity):
        if x.model is None or x.model._dev is None:
            return x.name[:3] + "[]"
        s = f"U:{np.linalg.norm(x.model.sample):6.2f}"
        if hasFeatures(x):
            s += f" β:{np.linalg.norm(x.model.beta):3.2f}"
            if x.lambda_beta_sample:
                s += f" λ={x.lambda_beta:1.1f}"
        return f"{x.name[:3]}[{s}]"
    s = f"α={x.model.alpha:2.1f}"
    if hasFeatures(x) and x.model.beta is not None and len(x.model.beta):
        s += f" β:{np.linalg.norm(x.model.beta):2.1f}"
    return f"{x.name[:4]}[{s}]"


class RelationModel:
    """RelationModel (RelationData.jl:107-119)"""

    def __init__(self, alpha=1.0, lambda_beta=1.0):
        self.alpha_sample = False
        self.alpha_nu0 = 2.0
        self.alpha_lambda0 = 1.0
        self.lambda_beta = float(lambda_beta)
        self.alpha = float(alpha)
        self.beta = np.zeros(0)
        self.mean_value = 0.0


class RelationTemp:
    def __init__(self):
        self.linear_values = None
        self.FF = None


class TestVec:
    """Relation.test_vec: the held-out rows of the table (ids 1-based)."""

    def __init__(self, ids, values, names):
        self.ids = np.asarray(ids)
        self.values = np.asarray(values, dtype=np.float64)
        self.names = list(names)

    def __len__(self):
        return len(self.values)

    @property
    def shape(self):
        return (len(self.values), self.ids.shape[1] + 1)

    def to_frame(self, **extra):
        import pandas as pd
        d = {self.names[k]: self.ids[:, k] for k in range(self.ids.shape[1])}
        d[self.names[-1]] = self.values
        d.update(extra)
        return pd.DataFrame(d)


def _table_from_sparse(M):
    """findnz(::SparseMatrixCSC) order: column-major (RelationData.jl:165-171, 299-305)"""
    csc = M.tocsc(copy=True)
    csc.sum_duplicates()
    csc.sort_indices()
    cols = np.repeat(np.arange(csc.shape[1], dtype=np.int64), np.diff(csc.indptr)) + 1
    rows = csc.indices.astype(np.int64) + 1
    return np.stack([rows, cols], axis=1), np.asarray(csc.data, dtype=np.float64)


class Relation:
    """Relation(data, name, entities=[]; class_cut=0.0, dims=...) (RelationData.jl:128-171).

    data: IndexedDF | pandas DataFrame / dict / (ids, values) table | scipy sparse matrix."""

    def __init__(self, data, name, entities=None, class_cut=0.0, alpha=1.0, dims=None):
        entities = list(entities) if entities is not None else []
        self.F = None
        self.name = str(name)
        self.class_cut = float(class_cut)
        self.model = RelationModel(alpha)
        self.temp = RelationTemp()
        self.test_F = None
        if isinstance(data, IndexedDF):
            self.data = data
            self.entities = entities
        else:
            if hasattr(data, "tocsc"):
                if len(entities) != 2:
                    raise ArgumentError("For matrix relation the number of entities has to be 2.")
                ids, vals = _table_from_sparse(data)
                names = ["E1", "E2", "values"]
                dims = [int(data.shape[0]), int(data.shape[1])]
            else:
                ids, vals, names = _split_table(data)
                if dims is None:
                    dims = [int(ids[:, i].max()) if len(ids) else 0 for i in range(ids.shape[1])]
                dims = [int(d) for d in dims]
            if entities:
                if ids.shape[1] != len(entities):
                    raise ArgumentError(f"data has {ids.shape[1] + 1} columns but needs to have {len(entities) + 1} "
                                        "which is number of entities + 1")
                for i, en in enumerate(entities):
                    if en.count == 0:
                        en.count = dims[i]
                    elif en.count > dims[i]:
                        dims[i] = en.count
                    elif en.count < dims[i]:
                        raise ArgumentError(f"Entity {en.name} has smaller count {en.count} than the largest id in the data "
                                            f"{dims[i]}. Set entity.count manually before creating the relation.")
            self.data = IndexedDF((ids, vals), dims, names=names)
            self.entities = entities
        self.test_vec = TestVec(self.data.ids[:0, :], self.data.values[:0], self.data.names)
        self.test_label = np.zeros(0, dtype=bool)
        self._dev = None          # engine.RelationState

    def size(self, d=None):
        return self.data.size(d)

    def __repr__(self):
        a = "sample" if self.model.alpha_sample else f"{self.model.alpha:.2f}"
        s = (f"[Relation] {self.name}: {'--'.join(e.name for e in self.entities)}, #known = {numData(self)}, "
             f"#test = {numTest(self)}, α = {a}")
        if hasFeatures(self):
            s += f", #feat = {feat.feature_shape(self.F)[1]}"
        return s


def numData(r):
    return r.data.nnz()


def numTest(r):
    return len(r.test_vec)


def setPrecision(r, precision):
    r.model.alpha = float(precision)


def assignToTest(r, test, rng=None):
    """assignToTest!(r, ntest::Int) / assignToTest!(r, test_id::Vector) (RelationData.jl:191-212); ids 1-based"""
    if np.isscalar(test):
        rng = rng if rng is not None else np.random.default_rng()
        test_id = rng.choice(r.data.nnz(), size=int(test), replace=False) + 1
    else:
        test_id = np.asarray(test, dtype=np.int64)
    rows0 = test_id - 1
    r.test_vec = TestVec(r.data.ids[rows0, :], r.data.values[rows0], r.data.names)
    r.data = r.data.removeSamples(test_id)
    r.test_label = r.test_vec.values < r.class_cut
    if hasFeatures(r):
        r.test_F = feat.take_rows(r.F, rows0)
        train = np.ones(feat.feature_shape(r.F)[0], dtype=bool)
        train[rows0] = False
        r.F = feat.subset_rows(r.F, train)
    r._dev = None
    return None


def setTest(r, test, test_feat=None):
    """setTest!(r, test_df[, test_feat]) / setTest!(r, test_mat::SparseMatrixCSC) (RelationData.jl:214-252)"""
    if hasattr(test, "tocsc"):
        if hasFeatures(r):
            raise ArgumentError("Cannot add test set using SparseMatrixCSC when relation has features. Use DataFrame instead.")
        if r.data.ids.shape[1] != 2:
            raise ArgumentError("Relation must have 2 entities if using SparseMatrixCSC for test set.")
        ids, vals = _table_from_sparse(test)
    else:
        ids, vals, _ = _split_table(test)
        if hasFeatures(r) and test_feat is None:
            raise ArgumentError("Relation has features, please supply features with test data:\nsetTest(rel, test_df, test_features")
        if hasFeatures(r) and feat.feature_shape(r.F)[1] != feat.feature_shape(test_feat)[1]:
            raise ArgumentError("The test_feat must have the same number of columns as relation.F.")
        if hasFeatures(r) and feat.feature_shape(test_feat)[0] != len(vals):
            raise ArgumentError("The test_feat must have the same number of rows as test_df.")
        if ids.shape[1] + 1 != r.data.ids.shape[1] + 1:
            raise ArgumentError("The number of columns in test_df must be the same as in relation.data.df.")
    r.test_vec = TestVec(ids, vals, r.data.names)
    r.test_label = r.test_vec.values < r.class_cut
    if hasFeatures(r):
        r.test_F = test_feat
    r._dev = None
    return None


class RelationData:
    """RelationData (RelationData.jl:254-311).

    RelationData()                         empty
    RelationData(relation)                 one relation with entities already attached (:307-311)
    RelationData(M; feat1, feat2, ...)     two-entity matrix relation from a scipy sparse matrix or an IndexedDF (:260-274, 293-305)
    RelationData(table; rname, ...)        N-mode relation from a table, one entity per id column (:276-286)
    """

    def __init__(self, data=None, feat1=None, feat2=None, entity1="E1", entity2="E2", relation="Rel", ntest=0,
                 class_cut=math.log10(200), alpha=5.0, alpha_sample=False, lambda_beta=1.0, rname="R1"):
        self.entities = []
        self.relations = []
        if data is None:
            return
        if isinstance(data, Relation):
            addRelation(self, data)
            return
        if hasattr(data, "tocsc"):
            ids, vals = _table_from_sparse(data)
            data = IndexedDF((ids, vals), [int(data.shape[0]), int(data.shape[1])], names=["row", "col", "value"])
        if isinstance(data, IndexedDF):
            if len(data.dims) != 2:
                raise ArgumentError("RelationData(::IndexedDF) builds a two-entity relation")
            r = Relation(data, relation, [], class_cut, 1.0 if alpha_sample else alpha)
            r.model.alpha_sample = bool(alpha_sample)
            e1 = Entity(entity1, F=feat1, lambda_beta=lambda_beta)
            e2 = Entity(entity2, F=feat2, lambda_beta=lambda_beta)
            e1.relations, e1.count = [r], r.size(1)
            e2.relations, e2.count = [r], r.size(2)
            if not feat.isempty(feat1) and feat.feature_shape(feat1)[0] != r.size(1):
                raise ArgumentError(f"Number of rows in feat1 {feat.feature_shape(feat1)[0]} must equal number of rows in the relation {r.size(1)}")
            if not feat.isempty(feat2) and feat.feature_shape(feat2)[0] != r.size(2):
                raise ArgumentError(f"Number of rows in feat2 {feat.feature_shape(feat2)[0]} must equal number of columns in the relation {r.size(2)}")
            r.entities = [e1, e2]
            self.entities = [e1, e2]
            self.relations = [r]
            if ntest:
                assignToTest(r, int(ntest))
            return
        # generic table: one entity per id column, named after the column
        ids, vals, names = _split_table(data)
        dims = [int(ids[:, i].max()) for i in range(ids.shape[1])]
        names = names or [f"E{i + 1}" for i in range(ids.shape[1])] + ["value"]
        idf = IndexedDF((ids, vals), dims, names=names)
        r = Relation(idf, rname, [], class_cut, alpha)
        self.relations.append(r)
        for d in range(len(dims)):
            en = Entity(names[d])
            en.relations, en.count = [r], idf.size(d + 1)
            self.entities.append(en)
            r.entities.append(en)

    def __repr__(self):
        out = ["[Relations]"]
        for r in self.relations:
            a = "sample" if r.model.alpha_sample else f"{r.model.alpha:.2f}"
            s = f"{r.name:>10s}: {'--'.join(e.name for e in r.entities)}, #known = {numData(r)}, #test = {numTest(r)}, α = {a}"
            if hasFeatures(r):
                s += f", #feat = {feat.feature_shape(r.F)[1]}"
            out.append(s)
        out.append("[Entities]")
        for en in self.entities:
            out.append(f"{en.name:>10s}: " + repr(en).split(": ", 1)[1])
        return "\n".join(out)


def addRelation(rd, r):
    """addRelation!(rd, r) (RelationData.jl:387-409).

    The reference registers r on an entity only when `! any(en.relations .!= r)` (:404), which drops every relation
    after the first on a shared entity although sample_user2 (sampling.jl:270-283) sums over all of them; the evident
    intent (register unless already present) is implemented here -- see DESIGN.md note N1."""
    if len(r.size()) != len(r.entities):
        raise ArgumentError(f"Relation has {len(r.entities)} entities but its data implies {r.size()}.")
    rd.relations.append(r)
    for i, en in enumerate(r.entities):
        if en.count == 0:
            en.count = r.size(i + 1)
        elif en.count != r.size(i + 1):
            raise ArgumentError(f"Entity {en.name} has {en.count} instances, relation {r.name} has data for {r.size(i + 1)}.")
        if not any(e is en for e in rd.entities):
            rd.entities.append(en)
        if not any(x is r for x in en.relations):
            en.relations.append(r)
    return None


def normalizeFeatures(entity):
    """normalizeFeatures!(entity) (RelationData.jl:450-454): unit column norms"""
    F = entity.F
    if hasattr(F, "tocsc"):
        import scipy.sparse as sp
        d = np.sqrt(np.asarray(F.multiply(F).sum(axis=0)).ravel())
        entity.F = (F @ sp.diags(1.0 / d)).tocsr()
    else:
        F = np.asarray(F, dtype=np.float64)
        entity.F = F / np.sqrt((F ** 2).sum(axis=0))[None, :]


def normalizeRows(entity):
    """normalizeRows!(entity) (RelationData.jl:456-459): unit row norms"""
    F = entity.F
    if hasattr(F, "tocsc"):
        import scipy.sparse as sp
        d = np.sqrt(np.asarray(F.multiply(F).sum(axis=1)).ravel())
        entity.F = (sp.diags(1.0 / d) @ F).tocsr()
    else:
        F = np.asarray(F, dtype=np.float64)
        entity.F = F / np.sqrt((F ** 2).sum(axis=1))[:, None]
