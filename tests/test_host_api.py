"""CPU suite: host-side behaviour of the drop-in VectorBase -- the bodies of the
reference's tests/test_vectorbase.py (/root/reference) that need no arithmetic,
transcribed (`pytest.mark.asyncio` -> asyncio.run, line 59's PEP 695 alias -> a
plain dict), plus the "fail loudly without a GPU" contract.  The lookups of
that file are in tests/test_gpu_parity.py (they need the device)."""

import asyncio

import numpy as np
import pytest

from tests.fakes import CachingEmbeddingModel, NamedModel, NullModel, create_test_embedding_model
from typeagent_py_amd import (
    DEFAULT_MIN_SCORE,
    ScoredInt,
    TextEmbeddingIndexSettings,
    VectorBase,
    cosine_to_score,
    get_default_min_score,
)


def make_vector_base() -> VectorBase:
    return VectorBase(TextEmbeddingIndexSettings(create_test_embedding_model()))


@pytest.fixture()
def vector_base() -> VectorBase:
    return make_vector_base()


@pytest.fixture()
def sample_embeddings() -> dict:
    return {
        "word1": np.array([0.1, 0.2, 0.3], dtype=np.float32),
        "word2": np.array([0.4, 0.5, 0.6], dtype=np.float32),
        "word3": np.array([0.7, 0.8, 0.9], dtype=np.float32),
    }


# --- reference tests/test_vectorbase.py:72-79
def test_add_embedding(vector_base, sample_embeddings):
    for key, embedding in sample_embeddings.items():
        vector_base.add_embedding(key, embedding)
    assert len(vector_base) == len(sample_embeddings)
    for i, (key, embedding) in enumerate(sample_embeddings.items()):
        np.testing.assert_array_equal(vector_base.serialize_embedding_at(i), embedding)


# --- :82-102
def test_add_embeddings(vector_base, sample_embeddings):
    keys = list(sample_embeddings.keys())
    for key, embedding in sample_embeddings.items():
        vector_base.add_embedding(key, embedding)
    bulk = make_vector_base()
    bulk.add_embeddings(keys, np.stack([sample_embeddings[k] for k in keys], axis=0))
    assert len(bulk) == len(vector_base)
    np.testing.assert_array_equal(bulk.serialize(), vector_base.serialize())
    assert isinstance(vector_base._model, CachingEmbeddingModel)
    assert isinstance(bulk._model, CachingEmbeddingModel)
    assert set(vector_base._model._cache.keys()) == set(bulk._model._cache.keys())
    for key in keys:
        np.testing.assert_array_equal(bulk._model._cache[key], vector_base._model._cache[key])


# --- :105-145
def test_add_key(vector_base, sample_embeddings):
    async def go():
        for key in sample_embeddings:
            await vector_base.add_key(key)

    asyncio.run(go())
    assert len(vector_base) == len(sample_embeddings)


def test_add_key_no_cache(vector_base, sample_embeddings):
    async def go():
        for key in sample_embeddings:
            await vector_base.add_key(key, cache=False)

    asyncio.run(go())
    assert len(vector_base) == len(sample_embeddings)
    assert vector_base._model._cache == {}, "Cache should remain empty when cache=False"


def test_add_keys(vector_base, sample_embeddings):
    out = asyncio.run(vector_base.add_keys(list(sample_embeddings.keys())))
    assert len(vector_base) == len(sample_embeddings)
    assert out.shape == (3, 3)
    assert asyncio.run(vector_base.add_keys([])) is None


def test_add_keys_no_cache(vector_base, sample_embeddings):
    asyncio.run(vector_base.add_keys(list(sample_embeddings.keys()), cache=False))
    assert len(vector_base) == len(sample_embeddings)
    assert vector_base._model._cache == {}


# --- :162-206
def test_clear(vector_base, sample_embeddings):
    for key, embedding in sample_embeddings.items():
        vector_base.add_embedding(key, embedding)
    assert len(vector_base) == 3
    vector_base.clear()
    assert len(vector_base) == 0
    assert vector_base.serialize().shape == (0, 3)  # clear keeps D


def test_serialize_deserialize(vector_base, sample_embeddings):
    for key, embedding in sample_embeddings.items():
        vector_base.add_embedding(key, embedding)
    serialized = vector_base.serialize()
    other = make_vector_base()
    other.deserialize(serialized)
    assert len(other) == len(vector_base)
    for i in range(len(vector_base)):
        np.testing.assert_array_equal(other.serialize_embedding_at(i), vector_base.serialize_embedding_at(i))
    assert other.serialize() is serialized  # adopted by reference (vectorbase.py:287)
    other.deserialize(None)
    assert len(other) == 0


def test_deserialize_shape_assert_and_empty():
    vb = make_vector_base()
    vb.deserialize(np.zeros((0,), dtype=np.float32))  # cannot learn D: just clears
    assert len(vb) == 0 and vb._embedding_size == 0
    vb.deserialize(np.ones((2, 4), dtype=np.float32))
    assert vb._embedding_size == 4 and len(vb) == 2
    with pytest.raises(AssertionError):
        vb.deserialize(np.ones((2, 5), dtype=np.float32))


def test_vectorbase_bool(vector_base):
    assert bool(vector_base) is True


def test_get_embedding_at(vector_base, sample_embeddings):
    for key, embedding in sample_embeddings.items():
        vector_base.add_embedding(key, embedding)
    for i, embedding in enumerate(sample_embeddings.values()):
        np.testing.assert_array_equal(vector_base.get_embedding_at(i), embedding)
    with pytest.raises(IndexError, match="Index 3 out of bounds for embedding index of size 3"):
        vector_base.get_embedding_at(len(sample_embeddings))
    assert vector_base.serialize_embedding_at(99) is None


# --- :255-277
def test_add_embedding_size_mismatch(vector_base):
    vector_base.add_embedding(None, np.array([0.1, 0.2, 0.3], dtype=np.float32))
    with pytest.raises(ValueError, match="Embedding size mismatch: expected 3, got 5"):
        vector_base.add_embedding(None, np.array([0.1, 0.2, 0.3, 0.4, 0.5], dtype=np.float32))


def test_add_embeddings_size_mismatch(vector_base):
    vector_base.add_embeddings(None, np.array([[0.1, 0.2, 0.3]], dtype=np.float32))
    with pytest.raises(ValueError, match="Embedding size mismatch"):
        vector_base.add_embeddings(None, np.array([[0.1, 0.2, 0.3, 0.4, 0.5]], dtype=np.float32))


def test_add_embeddings_wrong_ndim(vector_base):
    with pytest.raises(ValueError, match="Expected 2D embeddings array, got 1D"):
        vector_base.add_embeddings(None, np.array([0.1, 0.2, 0.3], dtype=np.float32))


def test_add_embedding_accepts_list_of_floats(vector_base):
    vector_base.add_embedding("k", [0.5, 0.25, 0.125])  # storage/memory/convthreads.py:81 does this
    assert vector_base.serialize().dtype == np.float32
    np.testing.assert_array_equal(vector_base.serialize()[0], np.array([0.5, 0.25, 0.125], dtype=np.float32))


# --- :280-325
@pytest.mark.parametrize(
    ("model_name", "expected"),
    [("text-embedding-3-large", 0.74), ("text-embedding-3-small", 0.73), ("text-embedding-ada-002", 0.93)],
)
def test_settings_known_model_default(model_name, expected):
    s = TextEmbeddingIndexSettings(embedding_model=NamedModel(model_name))
    assert s.min_score == expected
    assert s.max_matches is None


def test_settings_unknown_model_fallback():
    s = TextEmbeddingIndexSettings(embedding_model=NamedModel("custom-embedding-model"))
    assert s.min_score == DEFAULT_MIN_SCORE == 0.85
    assert s.max_matches is None and s.batch_size == 8
    assert get_default_min_score("nope") == 0.85


def test_settings_explicit_overrides_win():
    s = TextEmbeddingIndexSettings(embedding_model=NamedModel("text-embedding-3-large"), min_score=0.55, max_matches=7)
    assert s.min_score == 0.55 and s.max_matches == 7


def test_settings_invalid_max_matches_becomes_none():
    assert TextEmbeddingIndexSettings(embedding_model=NamedModel("x"), max_matches=0).max_matches is None
    assert TextEmbeddingIndexSettings(embedding_model=NamedModel("x"), batch_size=0).batch_size == 8


def test_cosine_to_score_matches_reference_map():
    c = np.array([-2.0, -1.0, 0.0, 0.5, 1.0, 3.0], dtype=np.float32)
    s = cosine_to_score(c)
    assert s.dtype == np.float32
    assert s.tolist() == [0.0, 0.0, 0.5, 0.75, 1.0, 1.0]


# --- growth / views -------------------------------------------------------------
def test_appends_do_not_alias_previous_views():
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    rng = np.random.default_rng(0)
    rows = rng.standard_normal((100, 8)).astype(np.float32)
    for r in rows[:37]:
        vb.add_embedding(None, r)
    snap = vb.serialize().copy()
    vb.add_embeddings(None, rows[37:])
    assert len(vb) == 100
    np.testing.assert_array_equal(vb.serialize()[:37], snap)
    np.testing.assert_array_equal(vb.serialize(), rows)
    assert vb.serialize().flags["C_CONTIGUOUS"]


# --- no GPU here: the product path must fail loudly, never fall back -----------------
def test_empty_index_lookups_need_no_device(vector_base):
    q = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    assert vector_base.fuzzy_lookup_embedding(q) == []
    assert vector_base.fuzzy_lookup_embedding_in_subset(q, [0, 1]) == []
    assert vector_base.fuzzy_lookup_embeddings(np.stack([q, q])) == [[], []]


def test_lookup_without_gpu_raises_instead_of_falling_back(vector_base, sample_embeddings):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    for key, embedding in sample_embeddings.items():
        vector_base.add_embedding(key, embedding)
    q = sample_embeddings["word1"]
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        vector_base.fuzzy_lookup_embedding(q)
    with pytest.raises(RuntimeError):
        vector_base.fuzzy_lookup_embedding_in_subset(q, [0])
    assert vector_base.fuzzy_lookup_embedding_in_subset(q, []) == []  # reference :214-215


def test_product_package_does_not_import_the_oracle():
    import os
    import re

    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "typeagent_py_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{f} imports the oracle"
                assert not re.search(r"np\.dot\(|argpartition\(|argsort\(|\.matmul\(|torch\.topk", text), f"{f} carries a host search path"


def test_scoredint_is_a_plain_dataclass():
    import dataclasses

    a = ScoredInt(3, 0.5)
    assert (a.item, a.score) == (3, 0.5) and a == ScoredInt(3, 0.5) and a != ScoredInt(3, 0.25)
    assert [f.name for f in dataclasses.fields(ScoredInt)] == ["item", "score"] and dataclasses.asdict(a) == {"item": 3, "score": 0.5}
    assert repr(a) == "ScoredInt(item=3, score=0.5)"
    a.score = 0.75  # mutable, like the reference's
    assert a.score == 0.75


def test_scoredint_matches_the_reference_dataclass():
    """Same fields, repr, equality and keyword construction as vectorbase.py:50-55 (needs /root/reference)."""
    import dataclasses

    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference checkout not present")
    ref = ref_loader.load_reference_vectorbase().ScoredInt
    ours, theirs = ScoredInt(item=7, score=0.25), ref(item=7, score=0.25)
    assert [f.name for f in dataclasses.fields(ours)] == [f.name for f in dataclasses.fields(theirs)]
    assert repr(ours) == repr(theirs) and dataclasses.astuple(ours) == dataclasses.astuple(theirs)
    assert (ours == ScoredInt(7, 0.25)) and (theirs == ref(7, 0.25)) and (ours != ScoredInt(8, 0.25))


def test_batch_lookup_as_arrays_and_shadow_env_knob_on_a_fake_engine(monkeypatch):
    from tests.fake_engine import FakeEngine
    from typeagent_py_amd import _native

    FakeEngine.instances = []
    monkeypatch.setattr(_native, "Engine", FakeEngine)
    monkeypatch.setenv("TYPEAGENT_VB_F32_SHADOW", "2")
    rng = np.random.default_rng(3)
    v = rng.standard_normal((300, 24)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    qs = v[[5, 9, 200]]
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    vb.add_embeddings(None, v)
    lists = vb.fuzzy_lookup_embeddings(qs, max_hits=4, min_score=0.6)
    o, s, c = vb.fuzzy_lookup_embeddings(qs, max_hits=4, min_score=0.6, as_arrays=True)
    assert FakeEngine.instances[0].options.get("f32_shadow") == 2
    assert o.shape == (3, 4) and [int(x) for x in o[:, 0]] == [5, 9, 200]
    for qi in range(3):
        assert [(r.item, r.score) for r in lists[qi]] == list(zip(o[qi, : c[qi]].tolist(), s[qi, : c[qi]].tolist()))
        assert all(isinstance(r.item, int) and isinstance(r.score, float) for r in lists[qi])
    with pytest.raises(ValueError):
        vb.fuzzy_lookup_embeddings(qs, max_hits=0, as_arrays=True)
    monkeypatch.setenv("TYPEAGENT_VB_F32_SHADOW", "0")
    vb0 = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    vb0.add_embeddings(None, v)
    vb0.fuzzy_lookup_embedding(qs[0])
    assert FakeEngine.instances[-1].options.get("f32_shadow") == 0


def test_hit_lists_built_in_c_equal_the_python_loop(monkeypatch):
    """`_tavb_pyhits.build` (csrc/tavb_pyhits.c) against the interpreter's loop: same classes, values, list shapes; bad shapes and
    classes without the two slots are refused."""
    import typeagent_py_amd.vectorbase as mod

    assert mod._tavb_pyhits is not None, "typeagent_py_amd/_tavb_pyhits.so is not built (make -C typeagent_py_amd/csrc)"
    rng = np.random.default_rng(3)
    o = rng.integers(0, 2**40, (37, 9)).astype(np.int64)
    s = rng.random((37, 9)).astype(np.float32)
    c = rng.integers(0, 10, 37).astype(np.int32)
    c[0], c[1] = 0, 9
    fast = mod._scored_lists(o, s, c, 9)
    monkeypatch.setattr(mod, "_tavb_pyhits", None)
    slow = mod._scored_lists(o, s, c, 9)
    assert fast == slow and [len(x) for x in fast] == c.tolist()
    assert all(type(h) is mod.ScoredInt and type(h.item) is int and type(h.score) is float for row in fast for h in row)
    assert fast[1][0].score == float(s[1, 0]) and fast[1][8].item == int(o[1, 8])
    h = fast[1][0]
    h.item = 5  # ordinary instances: writable, comparable, printable
    assert h == mod.ScoredInt(5, float(s[1, 0])) and repr(h).startswith("ScoredInt(item=5, score=")
    from typeagent_py_amd import _tavb_pyhits as ph

    with pytest.raises(ValueError):
        ph.build(mod.ScoredInt, o, s, c[:5], 9)
    with pytest.raises((TypeError, AttributeError)):
        ph.build(dict, o, s, c, 9)

    class Plain:  # no slots: refused, not corrupted
        item = 0
        score = 0.0

    with pytest.raises(TypeError):
        ph.build(Plain, o, s, c, 9)
    # strided / wrongly typed arrays take the Python loop
    monkeypatch.undo()
    assert mod._scored_lists(o[:, ::1], s, c.astype(np.int64).astype(np.int32), 9) == slow
