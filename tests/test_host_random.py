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


def _fresh(monkeypatch, n=50, d=8, seed=1, **kw):
    FakeEngine.instances = []
    monkeypatch.setattr(_native, "Engine", FakeEngine)
    rng = np.random.default_rng(seed)
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), **kw)
    rows = _unit(rng, n, d)
    vb.add_embeddings(None, rows)
    return vb, rows, rng


def _answers_for_current_rows(vb, q):
    live = np.asarray(vb.serialize()).copy()
    res = vb.fuzzy_lookup_embedding(q, max_hits=5, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(live, q), [r.item for r in res], [r.score for r in res], 5, 0.0)
    return res


@pytest.mark.parametrize("write", ["row", "element", "slice_of_view", "row_view", "imul", "ufunc_out", "copyto", "fill", "put", "putmask", "sort",
                                   "flat", "copyto_keywords", "take_out", "dot_out", "bulk_through_asarray", "bulk_through_torch"])
def test_writes_through_the_serialized_matrix_reach_the_device_mirror(monkeypatch, write):
    """serialize() hands out the live matrix like the reference (vectorbase.py:268-271), which always scores the live matrix (:176).
    Every write numpy can see -- on the array or on views derived from it -- must be answered for on the next lookup, whichever row
    it touches (the 32-row sampled fingerprint of round 2 missed single rows)."""
    vb, rows, rng = _fresh(monkeypatch)
    q = rows[17]
    first = _answers_for_current_rows(vb, q)
    assert first[0].item == 17
    m = vb.serialize()
    assert isinstance(m, np.ndarray) and m.dtype == np.float32 and m.shape == (50, 8)
    uploads = len(FakeEngine.instances[-1].uploads)
    new = _unit(rng, 1, 8)[0]
    if write == "row":
        m[17] = new
    elif write == "element":
        m[17, 3] = -m[17, 3] - 0.5
    elif write == "slice_of_view":
        m[10:20][7] = new
    elif write == "row_view":
        row = vb.get_embedding_at(17)
        row[:] = new
    elif write == "imul":
        m *= np.float32(-1.0)
    elif write == "ufunc_out":
        np.negative(m, out=m)
    elif write == "copyto":
        np.copyto(m, -np.asarray(m))
    elif write == "fill":
        m[17:18].fill(0.25)
    elif write == "put":
        m.put(np.arange(17 * 8, 18 * 8), new)
    elif write == "putmask":
        np.putmask(m, np.broadcast_to(np.arange(50)[:, None] == 17, m.shape), new)
    elif write == "sort":
        m.sort(axis=0)
    elif write == "flat":  # (round-3 advice: writers the view used to miss)
        m.flat[17 * 8 : 18 * 8] = new
    elif write == "copyto_keywords":
        np.copyto(dst=m, src=-np.asarray(m))
    elif write == "take_out":
        np.take(np.asarray(m).copy(), np.arange(50)[::-1], axis=0, out=m)
    elif write == "dot_out":
        np.dot(np.asarray(m).copy(), -np.eye(8, dtype=np.float32), out=m)
    elif write == "bulk_through_asarray":  # a base-class view: numpy tells the view nothing -- the fingerprint of a handed-out matrix does
        raw = np.asarray(m)
        raw *= np.float32(-1.0)
    elif write == "bulk_through_torch":
        import torch

        torch.from_numpy(np.asarray(m)).mul_(-1.0)
    after = _answers_for_current_rows(vb, q)
    assert len(FakeEngine.instances[-1].uploads) > uploads  # the mirror was refreshed
    assert [(r.item, r.score) for r in after] != [(r.item, r.score) for r in first]
    # reads do not dirty anything
    uploads = len(FakeEngine.instances[-1].uploads)
    _ = vb.serialize().sum(), vb.serialize()[3].copy(), np.linalg.norm(vb.serialize(), axis=1), vb.serialize() @ q, vb.serialize() * 2
    _answers_for_current_rows(vb, q)
    assert len(FakeEngine.instances[-1].uploads) == uploads
    assert type(vb.serialize() * 2) is np.ndarray  # arithmetic gives plain copies
    # arrays that only DERIVE from the matrix own their memory: editing them is nobody's business (round-3 advice: a private copy edited in
    # place used to trigger a re-upload of the whole corpus)
    m = vb.serialize()
    c = m.copy()
    c *= 2
    c[3] = 0
    row = vb.get_embedding_at(4).copy()
    row /= 3
    prod = np.dot(m, q)
    prod[0] = 7
    assert type(prod) is np.ndarray and type(np.sort(m, axis=0)) is np.ndarray
    _answers_for_current_rows(vb, q)
    assert len(FakeEngine.instances[-1].uploads) == uploads


def test_raw_writers_use_mark_dirty_and_adopted_matrices_follow_their_watch_mode(monkeypatch):
    vb, rows, rng = _fresh(monkeypatch)
    q = rows[5]
    _answers_for_current_rows(vb, q)
    raw = np.asarray(vb.serialize())  # a base-class view: numpy no longer tells the view about writes; the sampled fingerprint of a
    raw[5] = -raw[5]                  # handed-out matrix catches bulk edits (test above), not ONE row outside its 32-row sample:
    assert 5 not in np.unique(np.linspace(0, 49, num=32).astype(np.int64))
    stale = vb.fuzzy_lookup_embedding(q, max_hits=1, min_score=0.0)
    assert stale[0].item == 5  # the mirror is stale (documented residual) ...
    vb.mark_dirty()  # ... until the writer says so
    assert _answers_for_current_rows(vb, q)[0].item != 5

    # a matrix the CALLER owns (deserialize keeps it by reference, :287): the same object comes back, edits are watched by fingerprint
    for mode, single_row_seen in (("sampled", False), ("full", True)):
        FakeEngine.instances = []
        other = VectorBase(TextEmbeddingIndexSettings(NullModel()), verify_host=mode)
        data = _unit(rng, 200, 8)
        other.deserialize(data)
        assert other.serialize() is data
        q2 = data[101].copy()
        assert other.fuzzy_lookup_embedding(q2, max_hits=1)[0].item == 101
        data[101] = -data[101]  # one row, not among the 32 sampled ones (rows 0, 6, 12, ... of 200)
        got = other.fuzzy_lookup_embedding(q2, max_hits=1)[0].item
        assert (got != 101) == single_row_seen, mode
        data *= np.float32(-1.0)  # a bulk edit is seen by both
        res = other.fuzzy_lookup_embedding(q2, max_hits=3)
        vo.check_topk_parity(vo.scores_full(data, q2), [r.item for r in res], [r.score for r in res], 3, 0.0)
        # an append moves the index onto its own buffer (the reference's np.append copies too): edits made to `data` BEFORE the append are in
        data[7] = q2
        other.mark_dirty() if mode == "sampled" else None
        other.add_embedding(None, _unit(rng, 1, 8)[0])
        assert other.serialize() is not data and len(other) == 201
        assert other.fuzzy_lookup_embedding(q2, max_hits=1)[0].item == 7
    with pytest.raises(ValueError):
        VectorBase(TextEmbeddingIndexSettings(NullModel()), verify_host="sometimes")


def test_a_serialized_matrix_adopted_by_another_index_is_watched_by_both(monkeypatch):
    vb, rows, rng = _fresh(monkeypatch)
    m = vb.serialize()
    other = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    other.deserialize(m)  # e.g. a copy of an index built from its serialized form, by reference (:287)
    assert other.serialize() is m
    q = rows[30].copy()
    assert vb.fuzzy_lookup_embedding(q, max_hits=1)[0].item == 30 and other.fuzzy_lookup_embedding(q, max_hits=1)[0].item == 30
    m[30] = -m[30]
    m[2] = q
    assert vb.fuzzy_lookup_embedding(q, max_hits=1)[0].item == 2 and other.fuzzy_lookup_embedding(q, max_hits=1)[0].item == 2


def test_fallback_fingerprint_of_an_owned_matrix_is_amortised(monkeypatch):
    """A matrix this index owns reports writes through its tracking view; the fingerprint behind it (writers that go around numpy) costs as
    much as a small lookup.  Default: compared before every lookup once a view is out (a write that bypasses the tracker is seen by the very
    next lookup, as in the reference, which scores the live matrix).  `verify_host="lazy"` (opt-in): at most once per 64 lookups / 20 ms.
    A matrix the CALLER owns keeps the per-lookup check in every mode."""
    import time

    calls = []
    orig = VectorBase._fingerprint
    vb0, rows0, _ = _fresh(monkeypatch, n=400)
    q0 = rows0[5].copy()
    vb0.serialize()
    vb0.fuzzy_lookup_embedding(q0, max_hits=1)
    monkeypatch.setattr(VectorBase, "_fingerprint", lambda self: (calls.append(1), orig(self))[1])
    for _ in range(10):
        vb0.fuzzy_lookup_embedding(q0, max_hits=1)
    assert len(calls) == 10  # the default: every lookup
    raw0 = np.asarray(vb0.serialize())
    raw0 *= np.float32(-1.0)  # invisible to the tracker: seen by the next lookup all the same
    assert _answers_for_current_rows(vb0, q0)[0].item != 5
    monkeypatch.setattr(VectorBase, "_fingerprint", orig)

    vb, rows, rng = _fresh(monkeypatch, n=400)
    vb._verify_host = "lazy"
    q = rows[5].copy()
    vb.serialize()  # a view is out: the fallback watch is on
    vb.fuzzy_lookup_embedding(q, max_hits=1)
    calls.clear()
    monkeypatch.setattr(VectorBase, "_fingerprint", lambda self: (calls.append(1), orig(self))[1])
    t0 = time.monotonic()
    for _ in range(256):
        vb.fuzzy_lookup_embedding(q, max_hits=1)
    elapsed = time.monotonic() - t0
    assert len(calls) <= 256 // 64 + int(elapsed / 0.02) + 2, (len(calls), elapsed)
    # a bulk edit through a base-class view (invisible to the tracker) is still noticed: within 64 lookups or 20 ms
    raw = np.asarray(vb.serialize())
    raw *= np.float32(-1.0)
    time.sleep(0.03)
    assert _answers_for_current_rows(vb, q)[0].item != 5
    # ... and is not lost when an append regrows the buffer before the next check (the fingerprint is compared before the flag is dropped)
    vb2, rows2, _ = _fresh(monkeypatch, n=64)
    vb2._verify_host = "lazy"
    q2 = rows2[9].copy()
    vb2.serialize()
    vb2.fuzzy_lookup_embedding(q2, max_hits=1)
    raw2 = np.asarray(vb2.serialize())
    raw2 *= np.float32(-1.0)
    while len(vb2) < 200:  # grows past the buffer at least once
        vb2.add_embedding(None, _unit(rng, 1, 8)[0])
    assert _answers_for_current_rows(vb2, q2)[0].item != 9
    # the caller's own matrix: every lookup
    other = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    other.deserialize(_unit(rng, 100, 8))
    other.fuzzy_lookup_embedding(q, max_hits=1)
    calls.clear()
    for _ in range(10):
        other.fuzzy_lookup_embedding(q, max_hits=1)
    assert len(calls) == 10


def test_a_repeated_subset_keeps_its_row_list_on_the_device(monkeypatch):
    """`fuzzy_lookup_embedding_in_subset` (vectorbase.py:203-230) with the SAME list object again (the memory provider's scope list per query
    term, storage/memory/messageindex.py:173-183; tools/benchmark_vectorbase.py:133-163): the wrapped, range-checked row list is uploaded
    once.  Recognised by identity and content: a list edited in place, another list with the same content, or an index that has grown
    (negative ordinals wrap differently, the range check moves) is a new subset.  Answers are the oracle's every time."""
    vb, rows, rng = _fresh(monkeypatch, n=300)
    eng = None
    q = rows[11].copy()

    def check(subset, k=5):
        got = vb.fuzzy_lookup_embedding_in_subset(q, subset, max_hits=k, min_score=0.0)
        live = np.asarray(vb.serialize())
        ref = vo.lookup_in_subset(live, q, list(subset), k, 0.0)
        assert [r.item for r in got] == [i for i, _ in ref], subset[:8]
        return got

    subset = [5, 11, 250, -1, 11, 7]  # a duplicate and a negative (wrapping) ordinal, as numpy indexing takes them
    check(subset)
    eng = FakeEngine.instances[-1]
    assert eng.subset_uploads == 1
    for _ in range(5):
        check(subset)
    assert eng.subset_uploads == 1  # the same object with the same content: nothing re-uploaded
    subset[1] = 12  # edited in place: a new subset
    got = check(subset)
    assert eng.subset_uploads == 2 and 11 in [r.item for r in got] and got[0].item == 11  # (the duplicate at position 4 is still there)
    subset[4] = 13
    assert 11 not in [r.item for r in check(subset)] and eng.subset_uploads == 3
    check(list(subset))  # an equal list that is another object: uploaded again (identity is part of the key)
    assert eng.subset_uploads == 4
    arr = np.array([3, 11, 299, -300], dtype=np.int64)
    check(arr)
    check(arr)
    assert eng.subset_uploads == 5
    arr[0] = 4  # the caller's array edited in place
    check(arr)
    assert eng.subset_uploads == 6
    # the index grows: -1 is another row now
    sub2 = [-1, 0]
    before = check(sub2)
    vb.add_embedding(None, q)  # the new last row IS the query
    after = check(sub2)
    assert after[0].item == -1 and after[0].score > before[0].score and eng.subset_uploads == 8
    with pytest.raises(IndexError):
        vb.fuzzy_lookup_embedding_in_subset(q, [0, 301])
    # inputs that cannot be cached (a tuple) and the paged / all-survivors forms keep the plain path
    assert [r.item for r in vb.fuzzy_lookup_embedding_in_subset(q, (5, 11), max_hits=1)] == [11]
    assert len(vb.fuzzy_lookup_embedding_in_subset(q, sub2, max_hits=0, min_score=0.0)) == 2
    assert eng.subset_uploads == 8
    vb.clear()
    assert vb._subset_cache is None
