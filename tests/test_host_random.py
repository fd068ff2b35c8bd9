"""CPU suite: seeded random histories of ONE drop-in VectorBase over the numpy stand-in for the device engine (tests/fake_engine.py):
appends, deserialize / clear, in-place edits of the serialized matrix (with and without mark_dirty), interleaved with lookups that must
always answer for the CURRENT rows.  What is under test is the host bookkeeping of typeagent_py_amd/vectorbase.py: which rows it uploads
when, and that the device mirror never serves stale rows."""

import numpy as np
import pytest

from oracle import vectorbase_oracle as vo
from tests.fake_engine import FakeEngine
from tests.fakes import NullModel
from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase, _native


def _unit(rng, n, d):
    a = rng.standard_normal((n, d)).astype(np.float32)
    return a / np.linalg.norm(a, axis=1, keepdims=True)


@pytest.mark.parametrize("seed", range(60))
def test_random_history_on_a_fake_engine(monkeypatch, seed):
    FakeEngine.instances = []
    monkeypatch.setattr(_native, "Engine", FakeEngine)
    rng = np.random.default_rng(9000 + seed)
    d = int(rng.choice([3, 8, 32]))
    dtype = "fp16" if rng.random() < 0.3 else "fp32"
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=dtype)
    truth = np.zeros((0, d), dtype=np.float32)
    uploads_before, synced_rows = 0, -1
    for step in range(int(rng.integers(4, 12))):
        op = rng.choice(["add1", "addn", "addn", "deser", "clear", "edit_marked", "edit_bulk", "lookup_only"])
        n_before = len(truth)
        if op == "add1":
            row = _unit(rng, 1, d)[0]
            vb.add_embedding(None, row if rng.random() < 0.5 else row.tolist())
            truth = np.concatenate([truth, row[None, :]])
        elif op == "addn":
            rows = _unit(rng, int(rng.choice([1, 5, 64, 257])), d)
            vb.add_embeddings(None, rows)
            truth = np.concatenate([truth, rows])
        elif op == "deser":
            fresh = _unit(rng, int(rng.choice([0, 1, 30, 400])), d) if rng.random() < 0.8 else None
            vb.deserialize(fresh)
            truth = np.zeros((0, d), dtype=np.float32) if fresh is None else fresh.copy()
        elif op == "clear":
            vb.clear()
            truth = np.zeros((0, d), dtype=np.float32)
        elif op == "edit_marked" and len(truth):
            m = vb.serialize()
            i = int(rng.integers(len(truth)))
            m[i] = _unit(rng, 1, d)[0]
            truth[i] = m[i]
            vb.mark_dirty()
        elif op == "edit_bulk" and len(truth):
            m = vb.serialize()  # a whole-matrix edit (re-normalisation, sign flip): the sampled fingerprint notices it without mark_dirty
            m *= np.float32(-1.0)
            truth = -truth
        n = len(truth)
        assert len(vb) == n
        if n:
            np.testing.assert_array_equal(vb.serialize()[:n], truth)
        seen = truth.astype(np.float16).astype(np.float32) if dtype == "fp16" else truth
        q = truth[int(rng.integers(n))] if n and rng.random() < 0.6 else _unit(rng, 1, d)[0]
        k = int(rng.choice([1, 4, 10, 300]))
        ms = float(rng.choice([0.0, 0.5, 0.7]))
        res = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=ms)
        if n == 0:
            assert res == []
            synced_rows = -1
            continue
        vo.check_topk_parity(vo.scores_full(seen, q), [r.item for r in res], [r.score for r in res], k, ms)
        batch = vb.fuzzy_lookup_embeddings(np.stack([q, -q]), max_hits=min(k, 256), min_score=ms)
        vo.check_topk_parity(vo.scores_full(seen, -q), [r.item for r in batch[1]], [r.score for r in batch[1]], min(k, 256), ms)
        sub = rng.integers(0, n, size=int(rng.integers(1, min(n, 20) + 1))).tolist()
        got = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=k, min_score=ms)
        sub_a = np.asarray(sub, dtype=np.int64)
        vo.check_topk_parity(vo.scores_full(seen, q)[sub_a], [r.item for r in got], [r.score for r in got], k, ms, candidate_ordinals=sub_a)
        eng = FakeEngine.instances[-1]
        assert eng.rows == n
        if op in ("add1", "addn") and n_before > 0 and synced_rows == n_before:
            # an append after a lookup moves ONLY the new rows to the device (the reference re-copies the matrix, vectorbase.py:128, 145)
            assert len(eng.uploads) == uploads_before + 1 and eng.uploads[-1] == (n_before, n - n_before), (eng.uploads[-3:], n_before, n)
        elif op == "lookup_only":
            assert len(eng.uploads) == uploads_before  # nothing changed: nothing uploaded
        uploads_before, synced_rows = len(eng.uploads), n
