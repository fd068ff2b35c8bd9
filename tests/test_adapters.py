"""CPU suite for the host-side adapters around the hot path (SURVEY 8f "next" rows); the device
half (`lookup_texts_batched`) is in the GPU suite."""

import numpy as np
import pytest

from tests.fakes import NullModel
from typeagent_py_amd import ScoredInt, TextEmbeddingIndexSettings, VectorBase
from typeagent_py_amd.adapters import best_score_per_message, load_embeddings_bin, load_sqlite_embeddings


def test_best_score_per_message_matches_reference_aggregation():
    # rows 0,1 -> message 7; row 2 -> message 3; row 3 -> message 9
    hits = [ScoredInt(1, 0.9), ScoredInt(2, 0.8), ScoredInt(0, 0.95), ScoredInt(3, 0.8)]
    out = best_score_per_message(hits, [7, 7, 3, 9])
    assert [(h.item, h.score) for h in out] == [(7, 0.95), (3, 0.8), (9, 0.8)]  # stable among equal scores
    assert [(h.item, h.score) for h in best_score_per_message(hits, [7, 7, 3, 9], max_matches=1)] == [(7, 0.95)]
    only = best_score_per_message(hits, lambda r: [7, 7, 3, 9][r], accept=lambda m: m in {3, 9})
    assert [h.item for h in only] == [3, 9]


def test_load_embeddings_bin_streams_both_blocks(tmp_path):
    rng = np.random.default_rng(0)
    related = rng.standard_normal((37, 16)).astype("<f4")
    messages = rng.standard_normal((11, 16)).astype("<f4")
    path = tmp_path / "x_embeddings.bin"
    with open(path, "wb") as f:  # layout of knowpro/serialization.py:84-98
        f.write(related.tobytes())
        f.write(messages.tobytes())
    rv = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    mv = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    assert load_embeddings_bin(str(path), 16, 37, 11, rv, mv, chunk_rows=8) == (37, 11)
    np.testing.assert_array_equal(rv.serialize(), related)
    np.testing.assert_array_equal(mv.serialize(), messages)
    with pytest.raises(ValueError):
        load_embeddings_bin(str(path), 16, 37, 12, rv, mv)


def test_load_sqlite_embeddings_mirrors_the_reference_reload(tmp_path):
    import sqlite3

    rng = np.random.default_rng(1)
    db = sqlite3.connect(str(tmp_path / "c.db"))
    # the two tables the reference reloads from (storage/sqlite/schema.py:71-81, 131-136), reduced to the columns used
    db.execute("CREATE TABLE MessageTextIndex (msg_id INTEGER, chunk_ordinal INTEGER, embedding BLOB NOT NULL, index_position INTEGER)")
    db.execute("CREATE TABLE RelatedTermsFuzzy (term TEXT PRIMARY KEY, term_embedding BLOB NOT NULL)")
    msgs = rng.standard_normal((23, 12)).astype(np.float32)
    for i, row in enumerate(msgs):
        db.execute("INSERT INTO MessageTextIndex VALUES (?, ?, ?, ?)", (i // 2, i % 2, row.tobytes(), i))
    terms = {"pear": 2, "apple": 0, "zebra": 3, "mango": 1}
    tvecs = rng.standard_normal((4, 12)).astype(np.float32)
    for term, i in terms.items():
        db.execute("INSERT INTO RelatedTermsFuzzy VALUES (?, ?)", (term, tvecs[i].tobytes()))
    db.commit()
    mv = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    assert load_sqlite_embeddings(db, mv, fetch_rows=5) == []
    np.testing.assert_array_equal(mv.serialize(), msgs)
    tv = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    keys = load_sqlite_embeddings(db, tv, table="RelatedTermsFuzzy", column="term_embedding", order_by="term", key_column="term")
    assert keys == sorted(terms)  # ORDER BY term: row order != insertion order
    np.testing.assert_array_equal(tv.serialize(), np.stack([tvecs[terms[k]] for k in keys]))
    with pytest.raises(ValueError, match="Embedding size mismatch"):  # a 12-wide table into a 12-wide index is fine, 5-wide rows are not
        tv.add_embeddings(None, np.zeros((1, 5), np.float32))
    with pytest.raises(ValueError):
        load_sqlite_embeddings(db, mv, table="x; DROP TABLE y")


def test_row_to_message_map_that_grows_with_the_index_is_resnapshotted(monkeypatch):
    """A caller that keeps ONE list and appends to it as chunks are added (round-2 advice): the adapters used to key the device map on the
    object's identity alone and then raise 'the row -> message map covers L rows, the index has L+n'; an int64 array edited in place
    was aliased by the index.  Now: a snapshot (copy), retaken whenever the length changed or the index outgrew it."""
    from oracle import messages_oracle as mo
    from tests.fake_engine import FakeEngine
    from typeagent_py_amd import _native
    from typeagent_py_amd.adapters import lookup_messages_by_embedding, lookup_messages_in_subset

    FakeEngine.instances = []
    monkeypatch.setattr(_native, "Engine", FakeEngine)
    rng = np.random.default_rng(3)
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()))
    rows = rng.standard_normal((40, 8)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    row_to_msg: list[int] = []

    def check(q):
        got = lookup_messages_by_embedding(vb, q, row_to_msg, 6, 0.0)
        hits = [(h.item, h.score) for h in vb.fuzzy_lookup_embedding(q, max_hits=6, min_score=0.0)]
        assert [(g.item, g.score) for g in got] == mo.sqlite_messages_from_hits(hits, row_to_msg, None, 6)
        sub = list(range(0, len(vb), 3))
        got = lookup_messages_in_subset(vb, q, sub, row_to_msg, 4, 0.0)
        hits = [(h.item, h.score) for h in vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=4, min_score=0.0)]
        assert [(g.item, g.score) for g in got] == mo.memory_messages_from_hits(hits, row_to_msg)[:4]

    vb.add_embeddings(None, rows[:25])
    row_to_msg.extend(i // 2 for i in range(25))
    check(rows[3])
    vb.add_embeddings(None, rows[25:])  # the index grows, the SAME list grows with it
    row_to_msg.extend(100 + i for i in range(15))
    check(rows[30])
    assert FakeEngine.instances[-1].row_messages.tolist() == row_to_msg
    arr = np.asarray(row_to_msg, dtype=np.int64)
    vb.set_row_messages(arr)
    arr[:] = -5  # the caller's array is not the index's
    assert vb._row_messages.tolist() == row_to_msg
