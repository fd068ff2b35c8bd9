"""Drop-in proof in the build container (needs /root/reference; skipped on the GPU box): the reference's own VectorBase
CONSUMERS -- `knowpro/fuzzyindex.py`, `knowpro/textlocindex.py`, `storage/memory/convthreads.py`,
`storage/sqlite/messageindex.py`, `storage/sqlite/reltermsindex.py` -- and its `tools/benchmark_vectorbase.py` are
executed VERBATIM (oracle/ref_wrappers.py) once over the verbatim reference `VectorBase` and once over
`typeagent_py_amd.vectorbase.VectorBase`, and must behave the same.  There is no GPU here, so the new class computes through
the numpy stand-in engine of tests/fake_engine.py: what is proven is the class <-> consumer interface (names, signatures,
attribute reach-ins, return types, host bookkeeping); the numeric parity of the kernels is the GPU suite's job.
Also here: `install()` / `uninstall()` and `install_batched_lookup_terms()` against that throw-away `typeagent` package,
and the pin of oracle/messages_oracle.py to the verbatim sqlite provider.
"""

import asyncio
import contextlib
import io
import runpy
import sqlite3
import sys

import numpy as np
import pytest

from oracle import messages_oracle as mo
from oracle import ref_loader, ref_wrappers
from tests.fake_engine import FakeEngine
from tests.fakes import create_test_embedding_model
from tests.synth import make_corpus, make_queries

pytestmark = pytest.mark.skipif(not (ref_loader.reference_available() and ref_wrappers.consumers_available()),
                                reason="the verbatim reference is only present in the build container")

RELATED_TERMS_DDL = "CREATE TABLE RelatedTermsFuzzy (term TEXT NOT NULL PRIMARY KEY, term_embedding BLOB NOT NULL)"  # schema.py:131-136


def run(coro):
    return asyncio.run(coro)


@pytest.fixture
def both(monkeypatch):
    """(consumers over the verbatim reference class, consumers over the new class on the stand-in engine)"""
    import typeagent_py_amd.vectorbase as ours
    from typeagent_py_amd import _native

    monkeypatch.setattr(_native, "Engine", FakeEngine)
    ref_ns = ref_wrappers.load_consumers(ref_loader.load_reference_vectorbase())
    new_ns = ref_wrappers.load_consumers(ours)
    assert new_ns.fuzzyindex.VectorBase is ours.VectorBase and ref_ns.fuzzyindex.VectorBase is not ours.VectorBase
    return ref_ns, new_ns


def settings(ns, dim=16, **kw):
    # (the settings class comes from whichever vectorbase module the consumers bound)
    return ns.fuzzyindex.TextEmbeddingIndexSettings(embedding_model=create_test_embedding_model(dim), **kw)


def same_scored(a, b, key):
    assert [key(x) for x in a] == [key(x) for x in b]
    np.testing.assert_allclose([x.score for x in a], [x.score for x in b], atol=1e-6, rtol=0)


def test_embedding_index_and_text_location_index_behave_the_same(both):
    texts = [f"chunk number {i} about topic {i % 7}" for i in range(60)]
    outs = []
    for ns in both:
        idx = ns.fuzzyindex.EmbeddingIndex(settings(ns))
        assert run(idx.is_empty()) and len(idx) == 0
        run(idx.add_texts(texts[:40]))
        idx.push(run(idx._vector_base.get_embeddings(texts[40:])))
        assert len(idx) == 60 and run(idx.size()) == 60
        q = run(idx.get_embedding("chunk number 13 about topic 6"))
        near = idx.get_indexes_of_nearest(q, max_matches=7, min_score=0.0)
        sub = idx.get_indexes_of_nearest_in_subset(q, [5, 13, 13, 44, 2, -1], max_matches=4, min_score=0.0)
        pred = idx.get_indexes_of_nearest(q, max_matches=3, min_score=0.0, predicate=lambda i: i % 2 == 1)
        data = idx.serialize()
        idx2 = ns.fuzzyindex.EmbeddingIndex(settings(ns))
        idx2.deserialize(data)  # fuzzyindex.py:135-143 reads `_vector_base._embedding_size`
        with pytest.raises(AssertionError):
            idx2.deserialize(np.zeros((2, 5), dtype=np.float32))  # width mismatch once the size is known
        again = idx2.get_indexes_of_nearest(q, max_matches=7, min_score=0.0)
        np.testing.assert_array_equal(idx.get(3), data[3])
        idx.clear()
        assert len(idx) == 0
        # TextToTextLocationIndex
        TL = ns.interfaces.TextLocation
        tli = ns.textlocindex.TextToTextLocationIndex(settings(ns))
        run(tli.add_text_locations([(t, TL(i // 3, i % 3)) for i, t in enumerate(texts)]))
        hit = run(tli.lookup_text(texts[17], max_matches=5))  # default threshold 0.85 (textlocindex.py:108)
        hit_sub = run(tli.lookup_text_in_subset(texts[17], [1, 17, 18, 40], max_matches=3, threshold_score=0.0))
        by_emb = tli.lookup_by_embedding(q, 6, 0.0)
        ser = tli.serialize()
        tli2 = ns.textlocindex.TextToTextLocationIndex(settings(ns))
        tli2.deserialize(ser)
        by_emb2 = tli2.lookup_by_embedding(q, 6, 0.0)
        outs.append((near, sub, pred, again, hit, hit_sub, by_emb, by_emb2, run(tli2.size())))
    r, n = outs
    for a, b in zip(r[:4], n[:4]):
        same_scored(a, b, lambda x: x.item)
    for a, b in zip(r[4:8], n[4:8]):
        same_scored(a, b, lambda x: (x.text_location.message_ordinal, x.text_location.chunk_ordinal))
    assert r[8] == n[8] == 60 and r[4][0].text_location.message_ordinal == 17 // 3


def test_conversation_threads_behave_the_same(both):
    outs = []
    for ns in both:
        T = ns.interfaces.Thread
        ct = ns.convthreads.ConversationThreads(settings(ns, dim=3, min_score=0.0))
        for i, d in enumerate(["cooking pasta at home", "gpu kernels and matrix cores", "travel to iceland", "sourdough bread baking"]):
            run(ct.add_thread(T(d, [i])))
        first = run(ct.lookup_thread("gpu kernels and matrix cores", 3, 0.0))
        data = ct.serialize()
        assert all(isinstance(item["embedding"], list) for item in data["threads"])
        ct2 = ns.convthreads.ConversationThreads(settings(ns, dim=3, min_score=0.0))
        ct2.deserialize(data)  # add_embedding(description, list[float]) (convthreads.py:81)
        second = run(ct2.lookup_thread("sourdough bread baking"))
        run(ct2.build_index())
        third = run(ct2.lookup_thread("travel to iceland", 2))
        outs.append((first, second, third, len(ct2.vector_base)))
    r, n = outs
    for a, b in zip(r[:3], n[:3]):
        same_scored(a, b, lambda x: x.thread_ordinal)
    assert r[3] == n[3] == 4 and r[0][0].thread_ordinal == 1


class _Msg:
    def __init__(self, chunks):
        self.text_chunks = chunks


def _message_index(ns, db, dim):
    return ns.sqlite_messageindex.SqliteMessageTextIndex(db, ns.convsettings.MessageTextIndexSettings(settings(ns, dim=dim, min_score=0.0)))


def test_sqlite_message_index_behaves_the_same_and_pins_the_messages_oracle(both):
    v, _ = make_corpus(300, 24, 77)
    qs = make_queries(5, 24, 78)
    chunks_per_msg = [1 + (i % 3) for i in range(200)]  # 1..3 chunks per message
    msgs, pos = [], 0
    for c in chunks_per_msg:
        if pos + c > len(v):
            break
        msgs.append((_Msg([f"m{len(msgs)}c{j}" for j in range(c)]), list(v[pos : pos + c])))
        pos += c
    row_to_msg = [mi for mi, (m, _) in enumerate(msgs) for _ in m.text_chunks]
    outs = []
    for ns in both:
        db = sqlite3.connect(":memory:")
        db.execute(ref_wrappers.MESSAGE_TEXT_INDEX_DDL)
        idx = _message_index(ns, db, 24)
        run(idx.add_messages_starting_at_with_embeddings(0, [m for m, _ in msgs[:90]], [e for _, es in msgs[:90] for e in es]))
        run(idx.add_messages_starting_at_with_embeddings(90, [m for m, _ in msgs[90:]], [e for _, es in msgs[90:] for e in es]))
        idx_reloaded = _message_index(ns, db, 24)  # reload path: SELECT embedding FROM MessageTextIndex (:33-45)
        assert run(idx_reloaded.size()) == pos
        res = []
        for q in qs:
            res.append(run(idx_reloaded.lookup_by_embedding(q, 12, 0.0)))
            res.append(run(idx.lookup_by_embedding(q, None, 0.5)))
            res.append(run(idx.lookup_in_subset_by_embedding(q, list(range(0, len(msgs), 2)), 9, 0.0)))
        outs.append(res)
        if ns is both[0]:
            # pin the oracle restatement to the verbatim provider over the verbatim VectorBase
            vb = idx._vectorbase
            look = lambda e, k, t: [(s.item, s.score) for s in vb.fuzzy_lookup_embedding(e, max_hits=k, min_score=t)]
            for q in qs:
                for k, t, subset in ((12, 0.0, None), (None, 0.5, None), (9, 0.0, list(range(0, len(msgs), 2))), (40, 0.0, [3, 4, 5])):
                    want = (run(idx.lookup_by_embedding(q, k, t)) if subset is None else run(idx.lookup_in_subset_by_embedding(q, subset, k, t)))
                    got = mo.sqlite_lookup_by_embedding(look, q, row_to_msg, k, t, subset)
                    assert [(m.message_ordinal, m.score) for m in want] == got
    for a, b in zip(*outs):
        same_scored(a, b, lambda x: x.message_ordinal)


def test_sqlite_related_terms_and_the_batched_lookup_terms_patch(both, monkeypatch):
    terms = ["apple pie", "banana bread", "cherry tart", "apple tart", "banana split", "date square", "elderflower cordial"]
    outs = []
    for ns in both:
        db = sqlite3.connect(":memory:")
        db.execute(RELATED_TERMS_DDL)
        idx = ns.sqlite_reltermsindex.SqliteRelatedTermsFuzzy(db, settings(ns, dim=12, min_score=0.0, max_matches=4))
        run(idx.add_terms(terms))
        run(idx.add_terms(["apple pie", "fig roll"]))  # dedupe by _added_terms
        seq = run(idx.lookup_terms(["apple crumble", "banana loaf", "fig jam"], 3, 0.0))
        idx2 = ns.sqlite_reltermsindex.SqliteRelatedTermsFuzzy(db, settings(ns, dim=12, min_score=0.0, max_matches=4))  # ORDER BY term reload
        seq2 = run(idx2.lookup_terms(["apple crumble"]))
        outs.append((seq, seq2, run(idx.size())))
    (rs, rs2, rn), (ns_, ns2, nn) = outs
    for a, b in zip(rs + rs2, ns_ + ns2):
        same_scored([type("T", (), {"score": t.weight, "text": t.text}) for t in a], [type("T", (), {"score": t.weight, "text": t.text}) for t in b], lambda x: x.text)
    assert rn == nn == 8

    # the batched patch: same answers as the sequential loop it replaces
    import typeagent_py_amd.vectorbase as ours
    from typeagent_py_amd import adapters

    kept = ref_wrappers.load_consumers(ours, keep=True)
    try:
        report = adapters.install_batched_lookup_terms()
        assert report["patched"] == ["typeagent.storage.memory.reltermsindex.TermEmbeddingIndex", "typeagent.storage.sqlite.reltermsindex.SqliteRelatedTermsFuzzy"]
        assert report["skipped"] == {}
        db = sqlite3.connect(":memory:")
        db.execute(RELATED_TERMS_DDL)
        idx = kept.sqlite_reltermsindex.SqliteRelatedTermsFuzzy(db, settings(kept, dim=12, min_score=0.0, max_matches=4))
        run(idx.add_terms(terms))
        batched = run(idx.lookup_terms(["apple crumble", "banana loaf", "fig jam"], 3, 0.0))
        adapters.uninstall_batched_lookup_terms()
        sequential = run(idx.lookup_terms(["apple crumble", "banana loaf", "fig jam"], 3, 0.0))
        assert [[(t.text, t.weight) for t in ts] for ts in batched] == [[(t.text, t.weight) for t in ts] for ts in sequential]
    finally:
        adapters.uninstall_batched_lookup_terms()
        kept.cleanup()


def test_memory_message_index_behaves_the_same_and_pins_the_messages_oracle(both):
    """storage/memory/messageindex.py (the default in-memory provider's MessageTextIndex), executed through the PEP 695 source
    transform of oracle/ref_wrappers.py: add / lookup by embedding / lookup in subset / serialize -> deserialize, over both
    classes.  Note :173-183 hands MESSAGE ordinals to `TextToTextLocationIndex.lookup_in_subset_by_embedding`, which uses them as
    ROW ordinals of the VectorBase (textlocindex.py:164-177): with several chunks per message the subset form searches the wrong
    rows -- reproduced by both classes, restated by messages_oracle.memory_lookup_in_subset_by_embedding."""
    v, _ = make_corpus(300, 24, 87)
    qs = make_queries(5, 24, 88)
    chunks_per_msg = [1 + (i % 3) for i in range(200)]
    msgs, pos = [], 0
    for c in chunks_per_msg:
        if pos + c > len(v):
            break
        msgs.append((_Msg([f"m{len(msgs)}c{j}" for j in range(c)]), list(v[pos : pos + c])))
        pos += c
    row_to_msg = [mi for mi, (m, _) in enumerate(msgs) for _ in m.text_chunks]
    outs = []
    for ns in both:
        idx = ns.memory_messageindex.MessageTextIndex(ns.convsettings.MessageTextIndexSettings(settings(ns, dim=24, min_score=0.0)))
        assert run(idx.is_empty())
        run(idx.add_messages_starting_at_with_embeddings(0, [m for m, _ in msgs[:90]], [e for _, es in msgs[:90] for e in es]))
        run(idx.add_messages_starting_at_with_embeddings(90, [m for m, _ in msgs[90:]], [e for _, es in msgs[90:] for e in es]))
        assert run(idx.size()) == pos
        with pytest.raises(ValueError):
            run(idx.add_messages_starting_at_with_embeddings(len(msgs), [msgs[0][0]], []))
        data = run(idx.serialize())
        idx2 = ns.memory_messageindex.MessageTextIndex(ns.convsettings.MessageTextIndexSettings(settings(ns, dim=24, min_score=0.0)))
        run(idx2.deserialize(data))
        subset = list(range(0, len(msgs), 2))
        res = []
        for q in qs:
            res.append(run(idx.lookup_in_subset_by_embedding(q, subset, 9, 0.0)))
            res.append(run(idx2.lookup_in_subset_by_embedding(q, [3, 4, 5, 3], 40, 0.0)))
            res.append(run(idx.lookup_in_subset_by_embedding(q, subset, None, 0.5)))
            res.append(idx.to_scored_message_ordinals(idx.text_location_index.lookup_by_embedding(q, 12, 0.0)))
        outs.append(res)
        if ns is both[0]:
            vb = idx.text_location_index._embedding_index._vector_base  # (the verbatim class)
            look_sub = lambda e, rows, k, t: [(s.item, s.score) for s in vb.fuzzy_lookup_embedding_in_subset(e, rows, max_hits=k, min_score=t)]
            for q in qs:
                for k, t, sub in ((9, 0.0, subset), (40, 0.0, [3, 4, 5, 3]), (None, 0.5, subset)):
                    want = run(idx.lookup_in_subset_by_embedding(q, sub, k, t))
                    got = mo.memory_lookup_in_subset_by_embedding(look_sub, q, row_to_msg, sub, k, t)
                    assert [(m.message_ordinal, m.score) for m in want] == got
    for a, b in zip(*outs):
        same_scored(a, b, lambda x: x.message_ordinal)


def test_memory_term_embedding_index_and_the_batched_lookup_terms_patch(both):
    """storage/memory/reltermsindex.py: TermEmbeddingIndex (add / lookup_term / lookup_terms / serialize -> deserialize) and
    RelatedTermsIndex over both classes; then `install_batched_lookup_terms()` on the memory class -- the one the default
    in-memory provider uses (reltermsindex.py:320-332, called from :183-192): same answers as the sequential loop."""
    terms = ["apple pie", "banana bread", "cherry tart", "apple tart", "banana split", "date square", "elderflower cordial"]
    probes = ["apple crumble", "banana loaf", "fig jam"]
    outs = []
    for ns in both:
        M = ns.memory_reltermsindex
        idx = M.TermEmbeddingIndex(settings(ns, dim=12, min_score=0.0, max_matches=4))
        run(idx.add_terms(terms))
        run(idx.add_terms([]))
        with pytest.raises(ValueError):
            run(idx.add_terms_with_embeddings(["x"], []))
        seq = run(idx.lookup_terms(probes, 3, 0.0))
        one = run(idx.lookup_term("cherry cake"))  # defaults from the settings (max_matches=4, min_score=0.0)
        idx2 = M.TermEmbeddingIndex(settings(ns, dim=12, min_score=0.0, max_matches=4), idx.serialize())
        seq2 = run(idx2.lookup_terms(probes[:1]))
        rti = M.RelatedTermsIndex(ns.convsettings.RelatedTermIndexSettings(settings(ns, dim=12, min_score=0.0, max_matches=4)))
        run(rti.fuzzy_index.add_terms(terms))
        run(rti.aliases.add_related_term("pie", ns.interfaces.Term("tart", 0.9)))
        data = run(rti.serialize())
        rti2 = M.RelatedTermsIndex(ns.convsettings.RelatedTermIndexSettings(settings(ns, dim=12, min_score=0.0, max_matches=4)))
        run(rti2.deserialize(data))
        seq3 = run(rti2.fuzzy_index.lookup_terms(probes, 2, 0.0))
        outs.append((seq + [one] + seq2 + seq3, run(idx.size()), [t.text for t in run(rti2.aliases.lookup_term("pie"))]))
    (r, rn, ra), (n, nn, na) = outs
    as_scored = lambda ts: [type("T", (), {"score": t.weight, "text": t.text}) for t in ts]
    for a, b in zip(r, n):
        same_scored(as_scored(a), as_scored(b), lambda x: x.text)
    assert rn == nn == 7 and ra == na == ["tart"]

    import typeagent_py_amd.vectorbase as ours
    from typeagent_py_amd import adapters

    kept = ref_wrappers.load_consumers(ours, keep=True)
    try:
        report = adapters.install_batched_lookup_terms()
        assert "typeagent.storage.memory.reltermsindex.TermEmbeddingIndex" in report["patched"]
        idx = kept.memory_reltermsindex.TermEmbeddingIndex(settings(kept, dim=12, min_score=0.0, max_matches=4))
        run(idx.add_terms(terms))
        calls = []
        orig = ours.VectorBase.fuzzy_lookup_embeddings
        ours.VectorBase.fuzzy_lookup_embeddings = lambda self, *a, **kw: (calls.append(1), orig(self, *a, **kw))[1]
        try:
            batched = run(idx.lookup_terms(probes, 3, 0.0))
            batched_defaults = run(idx.lookup_terms(probes))
        finally:
            ours.VectorBase.fuzzy_lookup_embeddings = orig
        assert len(calls) == 2  # ONE device submission per lookup_terms call
        adapters.uninstall_batched_lookup_terms()
        sequential = run(idx.lookup_terms(probes, 3, 0.0))
        sequential_defaults = run(idx.lookup_terms(probes))
        flat = lambda res: [[(t.text, t.weight) for t in ts] for ts in res]
        assert flat(batched) == flat(sequential) and flat(batched_defaults) == flat(sequential_defaults)
        assert run(idx.lookup_terms([])) == []
    finally:
        adapters.uninstall_batched_lookup_terms()
        kept.cleanup()


def test_install_rebinds_the_reference_module_and_its_consumers_and_uninstall_restores(monkeypatch):
    import typeagent_py_amd
    import typeagent_py_amd.vectorbase as ours
    from typeagent_py_amd import _native

    monkeypatch.setattr(_native, "Engine", FakeEngine)
    ref_mod = ref_loader.load_reference_vectorbase()
    kept = ref_wrappers.load_consumers(ref_mod, keep=True)  # a `typeagent` package bound to the REFERENCE class
    try:
        assert kept.fuzzyindex.VectorBase is ref_mod.VectorBase
        touched = typeagent_py_amd.install()
        assert "typeagent.aitools.vectorbase" in touched and "typeagent.knowpro.fuzzyindex" in touched and "typeagent.storage.sqlite.messageindex" in touched
        assert ref_mod.VectorBase is ours.VectorBase and kept.fuzzyindex.VectorBase is ours.VectorBase
        assert kept.convthreads.VectorBase is ours.VectorBase and kept.textlocindex.ScoredInt is ours.ScoredInt
        idx = kept.fuzzyindex.EmbeddingIndex(kept.fuzzyindex.TextEmbeddingIndexSettings(embedding_model=create_test_embedding_model(8)))
        assert isinstance(idx._vector_base, ours.VectorBase)
        run(idx.add_texts(["one", "two", "three"]))
        assert idx.get_indexes_of_nearest(run(idx.get_embedding("two")), max_matches=1)[0].item == 1
        typeagent_py_amd.uninstall()
        assert ref_mod.VectorBase is not ours.VectorBase and kept.fuzzyindex.VectorBase is ref_mod.VectorBase
        assert kept.textlocindex.ScoredInt is ref_mod.ScoredInt
    finally:
        typeagent_py_amd.uninstall()
        kept.cleanup()
    assert "typeagent" not in sys.modules


def test_install_without_typeagent_registers_the_module_and_the_reference_benchmark_runs_unmodified(monkeypatch, capsys):
    """SURVEY 8b proof 3: tools/benchmark_vectorbase.py imports only `typeagent.aitools.{embeddings,vectorbase}` (:20-25);
    after `install()` it runs unmodified on the new class (here on the stand-in engine: the numbers mean nothing)."""
    import typeagent_py_amd
    import typeagent_py_amd.vectorbase as ours
    from typeagent_py_amd import _native

    monkeypatch.setattr(_native, "Engine", FakeEngine)
    assert "typeagent" not in sys.modules
    touched = typeagent_py_amd.install()
    try:
        assert touched == ["typeagent.aitools.vectorbase"] and sys.modules["typeagent.aitools.vectorbase"] is ours
        monkeypatch.setattr(sys, "argv", ["benchmark_vectorbase.py", "--rounds", "3", "--warmup-rounds", "1", "--dim", "64", "--subset-size", "100"])
        runpy.run_path(ref_loader.REFERENCE_ROOT + "/tools/benchmark_vectorbase.py", run_name="__main__")
        out = capsys.readouterr().out
        assert "fuzzy_lookup_embedding (10k vectors)" in out and "fuzzy_lookup_embedding_in_subset (100 of 10k)" in out
    finally:
        typeagent_py_amd.uninstall()
    assert "typeagent" not in sys.modules and "typeagent.aitools.vectorbase" not in sys.modules
