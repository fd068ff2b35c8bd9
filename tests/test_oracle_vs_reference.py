"""CPU suite, build container only: the numpy oracle against the VERBATIM reference
class (oracle/ref_loader.py executes /root/reference's vectorbase.py unmodified).
Skipped where /root/reference does not exist (the GPU box) -- there the committed
goldens (test_oracle_golden.py) carry the pin."""

import os

import numpy as np
import pytest

from oracle import ref_loader
from oracle import vectorbase_oracle as vo

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="/root/reference not mounted")


def _pairs(res):
    return [(int(r.item), float(r.score)) for r in res]


@pytest.mark.parametrize("n,d", [(1, 1), (7, 3), (300, 17), (1000, 384), (5000, 1536)])
@pytest.mark.parametrize("k", [None, 1, 10, 32, 50])
@pytest.mark.parametrize("min_score", [None, 0.0, 0.5, 0.53, 0.85, 1.0, 1.5])
def test_full_lookup_differential(n, d, k, min_score):
    rng = np.random.default_rng(n * 1000 + d)
    v = rng.standard_normal((n, d)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = v[rng.integers(n)] * 0.8 + 0.2 * rng.standard_normal(d).astype(np.float32)
    q = (q / np.linalg.norm(q)).astype(np.float32)
    ref = ref_loader.make_reference_vectorbase(v)
    assert vo.lookup(v, q, k, min_score) == _pairs(ref.fuzzy_lookup_embedding(q, max_hits=k, min_score=min_score))


def test_k_larger_than_n_and_zero_quirk_and_subset_and_predicate():
    rng = np.random.default_rng(1)
    v = rng.standard_normal((40, 6)).astype(np.float32)
    q = rng.standard_normal(6).astype(np.float32)
    ref = ref_loader.make_reference_vectorbase(v)
    for k in (41, 1000, 0):
        assert vo.lookup(v, q, k, 0.0) == _pairs(ref.fuzzy_lookup_embedding(q, max_hits=k, min_score=0.0))
    for sub in ([3, 3, 1, 0], [-1, 5], list(range(40)), [39]):
        assert vo.lookup_in_subset(v, q, sub, 5, 0.0) == _pairs(ref.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=5, min_score=0.0))
    with pytest.raises(IndexError):
        vo.lookup_in_subset(v, q, [40], 5, 0.0)
    with pytest.raises(IndexError):
        ref.fuzzy_lookup_embedding_in_subset(q, [40], max_hits=5, min_score=0.0)
    pred = lambda i: i % 3 == 1  # noqa: E731
    assert vo.lookup(v, q, 4, 0.2, pred) == _pairs(ref.fuzzy_lookup_embedding(q, max_hits=4, min_score=0.2, predicate=pred))


def test_cosine_to_score_and_settings_match():
    mod = ref_loader.load_reference_vectorbase()
    c = np.linspace(-1.5, 1.5, 101, dtype=np.float32)
    np.testing.assert_array_equal(vo.cosine_to_score(c), mod.cosine_to_score(c))
    from typeagent_py_amd import vectorbase as mine

    np.testing.assert_array_equal(mine.cosine_to_score(c), mod.cosine_to_score(c))
    assert mine.MODEL_DEFAULT_MIN_SCORES == mod.MODEL_DEFAULT_MIN_SCORES
    assert mine.DEFAULT_MIN_SCORE == mod.DEFAULT_MIN_SCORE
    for name in list(mod.MODEL_DEFAULT_MIN_SCORES) + ["other"]:
        assert mine.get_default_min_score(name) == mod.get_default_min_score(name)


def test_host_side_class_matches_reference_bookkeeping():
    """Same sequence of host-side calls on both classes -> same lengths, matrices and errors."""
    from tests.fakes import NullModel
    from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase

    mod = ref_loader.load_reference_vectorbase()
    ref = mod.VectorBase(mod.TextEmbeddingIndexSettings(embedding_model=NullModel()))
    mine = VectorBase(TextEmbeddingIndexSettings(embedding_model=NullModel()))
    rng = np.random.default_rng(2)
    a = rng.standard_normal((5, 4)).astype(np.float32)
    for vb in (ref, mine):
        assert len(vb) == 0 and bool(vb) and vb.serialize().shape == (0,)
        vb.add_embedding(None, a[0])
        vb.add_embedding(None, list(map(float, a[1])))
        vb.add_embeddings(None, a[2:])
        assert len(vb) == 5 and vb._embedding_size == 4
    np.testing.assert_array_equal(ref.serialize(), mine.serialize())
    for vb in (ref, mine):
        with pytest.raises(ValueError, match="Embedding size mismatch: expected 4, got 2"):
            vb.add_embedding(None, np.zeros(2, dtype=np.float32))
        with pytest.raises(ValueError, match="Expected 2D embeddings array, got 3D"):
            vb.add_embeddings(None, np.zeros((1, 1, 4), dtype=np.float32))
        with pytest.raises(IndexError, match="Index 5 out of bounds for embedding index of size 5"):
            vb.get_embedding_at(5)
        assert vb.serialize_embedding_at(-1) is None
        vb.clear()
        assert len(vb) == 0 and vb.serialize().shape == (0, 4)


def test_caching_embedding_model_makes_the_reference_calls():
    """`CachingEmbeddingModel` is registered as typeagent.aitools.embeddings' class by install(): embedders that count, log or bill their
    calls -- and batch-only ones -- must see the call pattern of the reference's class (aitools/embeddings.py:73-114).  The reference file is
    executed here with its two PEP 695 `type X = ...` alias statements rewritten as plain assignments (Python 3.10), nothing else touched."""
    import asyncio
    import re
    import types

    from typeagent_py_amd.embeddings import CachingEmbeddingModel

    path = os.path.join(ref_loader.REFERENCE_ROOT, "src", "typeagent", "aitools", "embeddings.py")
    src = open(path).read()
    new_src, n_changed = re.subn(r"(?m)^type (\w+) = ", r"\1 = ", src)
    assert n_changed == 2 and len(new_src.splitlines()) == len(src.splitlines())
    mod = types.ModuleType("ref_embeddings")
    exec(compile(new_src, path, "exec"), mod.__dict__)

    class Counting:
        model_name = "counting"

        def __init__(self):
            self.calls = []

        async def get_embedding_nocache(self, input):
            self.calls.append(("one", input))
            return np.full(3, float(len(input)), dtype=np.float32)

        async def get_embeddings_nocache(self, input):
            self.calls.append(("many", tuple(input)))
            return np.stack([np.full(3, float(len(s)), dtype=np.float32) for s in input])

    script = [("one", "a"), ("many", ["a", "bb", "bb", "ccc"]), ("one", "bb"), ("many", ["dddd"]), ("many", ["a", "dddd"]), ("one", "zz"),
              ("many", ["zz", "yy", "yy"]), ("add", "pre"), ("one", "pre"), ("many", ["pre", "q"])]
    logs, outs = [], []
    for cls in (mod.CachingEmbeddingModel, CachingEmbeddingModel):
        emb = Counting()
        model = cls(emb)
        out = []
        for op, arg in script:
            if op == "one":
                out.append(asyncio.run(model.get_embedding(arg)))
            elif op == "many":
                out.append(asyncio.run(model.get_embeddings(arg)))
            else:
                model.add_embedding(arg, np.full(3, 9.0, dtype=np.float32))
        logs.append(emb.calls)
        outs.append(out)
        assert sorted(model._cache) == sorted({"a", "bb", "ccc", "dddd", "zz", "yy", "pre", "q"})
    assert logs[0] == logs[1]
    for a, b in zip(*outs):
        assert a.dtype == b.dtype and a.shape == b.shape
        np.testing.assert_array_equal(a, b)
    for cls in (mod.CachingEmbeddingModel, CachingEmbeddingModel):
        with pytest.raises(ValueError):
            asyncio.run(cls(Counting()).get_embeddings([]))
