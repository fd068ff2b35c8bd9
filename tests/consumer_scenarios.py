"""Consumer-level scenarios: what the reference's VectorBase CONSUMERS return for fixed inputs.

Two drivers over the same deterministic inputs (seeded corpora of tests/synth.py, the hash embedder of tests/fakes.py):

  * `run_reference(ns)`  -- the reference's consumer files, executed VERBATIM (oracle/ref_wrappers.load_consumers) over a VectorBase
    class; `tests/golden/make_consumer_golden.py` runs it over the VERBATIM reference class in the build container and commits the
    answers as `tests/golden/consumer_golden.json`;
  * `run_replay(VectorBase, adapters)` -- the same calls through the product's consumer-side entry points (`typeagent_py_amd.adapters`,
    the class registered by `install()`), which is what runs on the GPU box (no /root/reference there): `-m gpu`
    tests/test_gpu_consumer_golden.py holds its output against the committed answers.

Every result is a list of `[id, score]` pairs: id = message ordinal / term text / thread ordinal / row ordinal / [message, chunk].

Reference lines replayed (/root/reference/src/typeagent):
  storage/sqlite/messageindex.py:182-257, 296-326   top-k chunk rows -> msg_id filter -> best score per message -> cut
  storage/memory/messageindex.py:139-207            lookup_messages / lookup_in_subset_by_embedding (message ordinals used as ROW ordinals)
  storage/memory/reltermsindex.py:313-337           lookup_term(s) -> Term(text, weight)
  storage/sqlite/reltermsindex.py:133-179, 259-271  reload ORDER BY term, lookup_term(s)
  storage/memory/convthreads.py:27-44               lookup_thread
  knowpro/fuzzyindex.py, knowpro/textlocindex.py:98-131   get_indexes_of_nearest(_in_subset), lookup_text (threshold 0.85 by default)
"""

from __future__ import annotations

import asyncio
import sqlite3

import numpy as np

from tests.fakes import create_test_embedding_model
from tests.synth import make_corpus, make_queries

MSG_DIM, MSG_ROWS, MSG_SEED = 64, 6000, 501
TERM_DIM, N_TERMS = 48, 3000
THREAD_DIM = 48

RELATED_TERMS_DDL = "CREATE TABLE RelatedTermsFuzzy (term TEXT NOT NULL PRIMARY KEY, term_embedding BLOB NOT NULL)"  # storage/sqlite/schema.py:131-136
MESSAGE_TEXT_INDEX_DDL = """
CREATE TABLE MessageTextIndex (
    msg_id INTEGER NOT NULL,
    chunk_ordinal INTEGER NOT NULL,
    embedding BLOB NOT NULL,
    index_position INTEGER
)
"""  # storage/sqlite/schema.py:71-81


def run(coro):
    return asyncio.run(coro)


# ---------------------------------------------------------------------------------------------------------------------
# inputs
# ---------------------------------------------------------------------------------------------------------------------
def message_inputs():
    """6000 chunk rows (seeded gaussians, D = 64) laid out as messages of 1..3 chunks; 4 queries next to rows of the corpus (so that
    thresholds above 0.5 keep something) and 2 unrelated ones."""
    v, _ = make_corpus(MSG_ROWS, MSG_DIM, MSG_SEED)
    chunks, pos = [], 0
    while pos < len(v):
        c = min(1 + (len(chunks) % 3), len(v) - pos)
        chunks.append(c)
        pos += c
    row_to_msg = np.repeat(np.arange(len(chunks)), chunks).astype(np.int64)
    rng = np.random.default_rng(MSG_SEED + 1)
    near = []
    for r in (17, 1234, 4321, 5999):
        q = v[r] + 0.8 * rng.standard_normal(MSG_DIM).astype(np.float32) / np.sqrt(MSG_DIM)
        near.append((q / np.linalg.norm(q)).astype(np.float32))
    queries = np.concatenate([np.stack(near), make_queries(2, MSG_DIM, MSG_SEED + 2)])
    n_msgs = len(chunks)
    subsets = {"even": list(range(0, n_msgs, 2)), "few": [3, 4, 5], "dups": [3, 4, 5, 3], "tail": list(range(n_msgs - 400, n_msgs))}
    return v, chunks, row_to_msg, queries, subsets


class _Msg:
    def __init__(self, chunks):
        self.text_chunks = chunks


def _messages(chunks):
    return [_Msg([f"m{i}c{j}" for j in range(c)]) for i, c in enumerate(chunks)]


_WORDS = ["apple", "banana", "cherry", "date", "elderflower", "fig", "grape", "hazelnut", "iceberg", "jasmine", "kiwi", "lemon", "mango", "nectarine", "olive",
          "papaya", "quince", "raspberry", "saffron", "tomato", "vanilla", "walnut", "yam", "zucchini"]
_KINDS = ["pie", "bread", "tart", "split", "square", "cordial", "crumble", "loaf", "jam", "roll", "soup", "salad", "cake"]


def term_inputs():
    terms = [f"{_WORDS[i % len(_WORDS)]} {_KINDS[(i // len(_WORDS)) % len(_KINDS)]} number {i}" for i in range(N_TERMS)]
    probes = [f"{_WORDS[(7 * i) % len(_WORDS)]} {_KINDS[(3 * i) % len(_KINDS)]} recipe {i}" for i in range(80)]
    probes[5] = terms[1234]  # an exact hit
    return terms, probes


def thread_inputs():
    descs = [f"thread about {_WORDS[i % len(_WORDS)]} {_KINDS[i % len(_KINDS)]} in episode {i}" for i in range(40)]
    probes = [descs[7], "thread about kiwi jam in episode 99", "gpu kernels and matrix cores"]
    return descs, probes


def text_inputs():
    texts = [f"chunk number {i} about topic {i % 7}" for i in range(240)]
    return texts


def _pairs(hits, key):
    """One result list of the golden file: {"hits": [[id, score], ...]}."""
    return {"hits": [[key(h), float(h.score)] for h in hits]}


# ---------------------------------------------------------------------------------------------------------------------
# the reference's consumers, verbatim
# ---------------------------------------------------------------------------------------------------------------------
def run_reference(ns) -> dict:
    """`ns` = oracle.ref_wrappers.load_consumers(vectorbase module): the verbatim consumer modules bound to that module's classes."""
    S = ns.fuzzyindex.TextEmbeddingIndexSettings

    def settings(dim, **kw):
        return S(embedding_model=create_test_embedding_model(dim), **kw)

    out: dict = {}
    v, chunks, row_to_msg, queries, subsets = message_inputs()
    msgs = _messages(chunks)
    # -- sqlite provider's message index (reloaded from its table, as at open time)
    db = sqlite3.connect(":memory:")
    db.execute(MESSAGE_TEXT_INDEX_DDL)
    mk = lambda: ns.sqlite_messageindex.SqliteMessageTextIndex(db, ns.convsettings.MessageTextIndexSettings(settings(MSG_DIM, min_score=0.0)))
    idx = mk()
    half = len(msgs) // 2
    rows_half = int(np.sum(chunks[:half]))
    run(idx.add_messages_starting_at_with_embeddings(0, msgs[:half], list(v[:rows_half])))
    run(idx.add_messages_starting_at_with_embeddings(half, msgs[half:], list(v[rows_half:])))
    idx = mk()
    assert run(idx.size()) == len(v)
    mo = lambda hits: _pairs(hits, lambda h: int(h.message_ordinal))
    res = []
    for q in queries:
        res.append(mo(run(idx.lookup_by_embedding(q, 25, 0.0))))
        res.append(mo(run(idx.lookup_by_embedding(q, None, 0.55))))
        res.append(mo(run(idx.lookup_by_embedding(q, 300, 0.6))))
        res.append(mo(run(idx.lookup_in_subset_by_embedding(q, subsets["even"], 25, 0.0))))
        res.append(mo(run(idx.lookup_in_subset_by_embedding(q, subsets["few"], 40, 0.0))))
        res.append(mo(run(idx.lookup_in_subset_by_embedding(q, subsets["tail"], 100, 0.5))))
    out["sqlite_messages"] = res
    # -- memory provider's message index
    midx = ns.memory_messageindex.MessageTextIndex(ns.convsettings.MessageTextIndexSettings(settings(MSG_DIM, min_score=0.0)))
    run(midx.add_messages_starting_at_with_embeddings(0, msgs[:half], list(v[:rows_half])))
    run(midx.add_messages_starting_at_with_embeddings(half, msgs[half:], list(v[rows_half:])))
    res = []
    for q in queries:
        res.append(mo(run(midx.lookup_in_subset_by_embedding(q, subsets["even"], 9, 0.0))))
        res.append(mo(run(midx.lookup_in_subset_by_embedding(q, subsets["dups"], 40, 0.0))))
        res.append(mo(run(midx.lookup_in_subset_by_embedding(q, subsets["tail"], None, 0.5))))
        res.append(mo(midx.to_scored_message_ordinals(midx.text_location_index.lookup_by_embedding(q, 12, 0.0))))
    out["memory_messages"] = res
    # -- related terms, memory provider: lookup_terms for T = 1 / 4 / 32 / 80 texts (sequential fuzzy_lookup loop, :320-332)
    terms, probes = term_inputs()
    tidx = ns.memory_reltermsindex.TermEmbeddingIndex(settings(TERM_DIM, min_score=0.85, max_matches=50))
    run(tidx.add_terms(terms))
    tt = lambda lists: [{"hits": [[t.text, float(t.weight)] for t in ts]} for ts in lists]
    out["memory_terms"] = {
        "T1": tt(run(tidx.lookup_terms(probes[:1]))),                 # the settings' defaults: k = 50 @ 0.85 (knowpro/convsettings.py:61-63)
        "T4": tt(run(tidx.lookup_terms(probes[1:5], 5, 0.0))),
        "T32": tt(run(tidx.lookup_terms(probes[:32]))),
        "T80": tt(run(tidx.lookup_terms(probes, 10, 0.9))),
        "one": tt([run(tidx.lookup_term(probes[5], 3, 0.0))]),
    }
    # -- related terms, sqlite provider: written, then reloaded ORDER BY term (:133-156)
    db2 = sqlite3.connect(":memory:")
    db2.execute(RELATED_TERMS_DDL)
    sidx = ns.sqlite_reltermsindex.SqliteRelatedTermsFuzzy(db2, settings(TERM_DIM, min_score=0.85, max_matches=50))
    run(sidx.add_terms(terms[:1000]))
    sidx = ns.sqlite_reltermsindex.SqliteRelatedTermsFuzzy(db2, settings(TERM_DIM, min_score=0.85, max_matches=50))
    out["sqlite_terms"] = {"T4": tt(run(sidx.lookup_terms(probes[1:5], 5, 0.0))), "T32": tt(run(sidx.lookup_terms(probes[:32])))}
    # -- conversation threads
    descs, tprobes = thread_inputs()
    T = ns.interfaces.Thread
    ct = ns.convthreads.ConversationThreads(settings(THREAD_DIM, min_score=0.7, max_matches=10))
    for i, d in enumerate(descs):
        run(ct.add_thread(T(d, [i])))
    th = lambda hits: _pairs(hits, lambda h: int(h.thread_ordinal))
    out["threads"] = [th(run(ct.lookup_thread(tprobes[0]))), th(run(ct.lookup_thread(tprobes[1], 3, 0.0))), th(run(ct.lookup_thread(tprobes[2], 10, 0.7)))]
    # -- EmbeddingIndex / TextToTextLocationIndex
    texts = text_inputs()
    eidx = ns.fuzzyindex.EmbeddingIndex(settings(32))
    run(eidx.add_texts(texts))
    q = run(eidx.get_embedding("chunk number 13 about topic 6"))
    it = lambda hits: _pairs(hits, lambda h: int(h.item))
    TL = ns.interfaces.TextLocation
    tli = ns.textlocindex.TextToTextLocationIndex(settings(32))
    run(tli.add_text_locations([(t, TL(i // 3, i % 3)) for i, t in enumerate(texts)]))
    loc = lambda hits: _pairs(hits, lambda h: [int(h.text_location.message_ordinal), int(h.text_location.chunk_ordinal)])
    out["embedding_index"] = {
        "nearest": it(eidx.get_indexes_of_nearest(q, max_matches=7, min_score=0.0)),
        "subset": it(eidx.get_indexes_of_nearest_in_subset(q, [5, 13, 13, 44, 2, -1], max_matches=4, min_score=0.0)),
        "predicate": it(eidx.get_indexes_of_nearest(q, max_matches=3, min_score=0.0, predicate=lambda i: i % 2 == 1)),
        "lookup_text": loc(run(tli.lookup_text(texts[17], max_matches=5))),  # default threshold 0.85 (textlocindex.py:108)
        "lookup_text_in_subset": loc(run(tli.lookup_text_in_subset(texts[17], [1, 17, 18, 40], max_matches=3, threshold_score=0.0))),
    }
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the same calls through the product's consumer-side entry points (runs on the GPU box)
# ---------------------------------------------------------------------------------------------------------------------
def run_replay(vb_module, adapters) -> dict:
    """`vb_module` = the module `install()` registered as `typeagent.aitools.vectorbase` (VectorBase, TextEmbeddingIndexSettings)."""
    VectorBase, S = vb_module.VectorBase, vb_module.TextEmbeddingIndexSettings

    def settings(dim, **kw):
        return S(embedding_model=create_test_embedding_model(dim), **kw)

    it = lambda hits: _pairs(hits, lambda h: int(h.item))
    out: dict = {}
    v, chunks, row_to_msg, queries, subsets = message_inputs()
    vb = VectorBase(settings(MSG_DIM, min_score=0.0))
    half_rows = int(np.sum(chunks[: len(chunks) // 2]))
    vb.add_embeddings(None, v[:half_rows])
    vb.add_embeddings(None, v[half_rows:])
    res = []
    for q in queries:  # storage/sqlite/messageindex.py:296-326
        res.append(it(adapters.lookup_messages_by_embedding(vb, q, row_to_msg, 25, 0.0)))
        res.append(it(adapters.lookup_messages_by_embedding(vb, q, row_to_msg, None, 0.55)))
        res.append(it(adapters.lookup_messages_by_embedding(vb, q, row_to_msg, 300, 0.6)))
        res.append(it(adapters.lookup_messages_by_embedding(vb, q, row_to_msg, 25, 0.0, accept=subsets["even"])))
        res.append(it(adapters.lookup_messages_by_embedding(vb, q, row_to_msg, 40, 0.0, accept=subsets["few"])))
        res.append(it(adapters.lookup_messages_by_embedding(vb, q, row_to_msg, 100, 0.5, accept=subsets["tail"])))
    out["sqlite_messages"] = res
    res = []
    for q in queries:  # storage/memory/messageindex.py:173-207: the caller's message ordinals are the ROW ordinals of the gather
        res.append(it(adapters.lookup_messages_in_subset(vb, q, subsets["even"], row_to_msg, 9, 0.0)))
        res.append(it(adapters.lookup_messages_in_subset(vb, q, subsets["dups"], row_to_msg, 40, 0.0)))
        res.append(it(adapters.lookup_messages_in_subset(vb, q, subsets["tail"], row_to_msg, None, 0.5)))
        res.append(it(adapters.lookup_messages_by_embedding(vb, q, row_to_msg, 12, 0.0)))
    out["memory_messages"] = res
    terms, probes = term_inputs()
    tvb = VectorBase(settings(TERM_DIM, min_score=0.85, max_matches=50))
    run(tvb.add_keys(terms))

    def lookup_terms(base, names, texts, max_hits=None, min_score=None):  # storage/memory/reltermsindex.py:320-337 as ONE submission
        lists = run(adapters.lookup_texts_batched(base, texts, max_hits, min_score))
        return [{"hits": [[names[m.item], float(m.score)] for m in ms if m.item < len(names)]} for ms in lists]

    out["memory_terms"] = {
        "T1": lookup_terms(tvb, terms, probes[:1]),
        "T4": lookup_terms(tvb, terms, probes[1:5], 5, 0.0),
        "T32": lookup_terms(tvb, terms, probes[:32]),
        "T80": lookup_terms(tvb, terms, probes, 10, 0.9),
        "one": [{"hits": [[terms[m.item], float(m.score)] for m in run(tvb.fuzzy_lookup(probes[5], 3, 0.0))]}],
    }
    # the sqlite provider reloads its rows ORDER BY term: the BLOB loader of the product does that read (adapters.load_sqlite_embeddings)
    db2 = sqlite3.connect(":memory:")
    db2.execute(RELATED_TERMS_DDL)
    emb = run(tvb.get_embeddings(terms[:1000]))
    db2.executemany("INSERT INTO RelatedTermsFuzzy (term, term_embedding) VALUES (?, ?)", [(t, np.asarray(e, dtype=np.float32).tobytes()) for t, e in zip(terms[:1000], emb)])
    svb = VectorBase(settings(TERM_DIM, min_score=0.85, max_matches=50))
    names = adapters.load_sqlite_embeddings(db2, svb, table="RelatedTermsFuzzy", column="term_embedding", order_by="term", key_column="term")
    out["sqlite_terms"] = {"T4": lookup_terms(svb, names, probes[1:5], 5, 0.0), "T32": lookup_terms(svb, names, probes[:32])}
    descs, tprobes = thread_inputs()
    cvb = VectorBase(settings(THREAD_DIM, min_score=0.7, max_matches=10))
    for d in descs:
        run(cvb.add_key(d, cache=False))  # storage/memory/convthreads.py:27-31
    out["threads"] = [it(run(cvb.fuzzy_lookup(tprobes[0], None, None))), it(run(cvb.fuzzy_lookup(tprobes[1], 3, 0.0))), it(run(cvb.fuzzy_lookup(tprobes[2], 10, 0.7)))]
    texts = text_inputs()
    evb = VectorBase(settings(32))
    run(evb.add_keys(texts))
    q = run(evb.get_embedding("chunk number 13 about topic 6"))
    q17 = run(evb.get_embedding(texts[17]))
    loc = lambda hits: {"hits": [[[h.item // 3, h.item % 3], float(h.score)] for h in hits]}
    out["embedding_index"] = {
        "nearest": it(evb.fuzzy_lookup_embedding(q, max_hits=7, min_score=0.0)),
        "subset": it(evb.fuzzy_lookup_embedding_in_subset(q, [5, 13, 13, 44, 2, -1], max_hits=4, min_score=0.0)),
        "predicate": it(evb.fuzzy_lookup_embedding(q, max_hits=3, min_score=0.0, predicate=lambda i: i % 2 == 1)),
        "lookup_text": loc(evb.fuzzy_lookup_embedding(q17, max_hits=5, min_score=0.85)),
        "lookup_text_in_subset": loc(evb.fuzzy_lookup_embedding_in_subset(q17, [1, 17, 18, 40], max_hits=3, min_score=0.0)),
    }
    return out


# ---------------------------------------------------------------------------------------------------------------------
# comparison
# ---------------------------------------------------------------------------------------------------------------------
TIE_EPS = 4 * 2.0 ** -24   # two float32 scores this close are a tie the reference orders by numpy's introselect (oracle.vectorbase_oracle.TIE_EPS)
SCORE_TOL = 1e-5           # north star: cosine scores within 1e-5


def compare_lists(got, want, where: str) -> int:
    """ids equal position by position and scores within SCORE_TOL; a different id at a position is accepted only when it trades places
    with another row of (near-)equal reference score -- a tie the reference itself orders arbitrarily -- or, at the END of a list that a
    `max_matches` cut truncated, replaces a row whose score ties with the score at the cut.  Returns the number of such positions."""
    assert len(got) == len(want), f"{where}: {len(got)} results, the reference returned {len(want)}"
    swapped = 0
    want_ids = [w[0] for w in want]
    for i, ((gi, gs), (wi, ws)) in enumerate(zip(got, want)):
        assert abs(gs - ws) <= SCORE_TOL, f"{where}[{i}]: score {gs} vs {ws}"
        if gi != wi:
            swapped += 1
            if gi in want_ids:
                j = want_ids.index(gi)
                assert abs(want[j][1] - ws) <= TIE_EPS, f"{where}[{i}]: {gi!r} instead of {wi!r} without a tie ({want[j][1]} vs {ws})"
            else:
                assert abs(want[-1][1] - ws) <= TIE_EPS, f"{where}[{i}]: {gi!r} is not in the reference's answer and {wi!r} does not tie with the cut"
    return swapped


def compare(got, want) -> tuple[int, int]:
    """Walks two structures of the golden file's shape ({"hits": [...]} leaves); returns (lists compared, positions that differed inside ties)."""
    n = swapped = 0

    def walk(g, w, where):
        nonlocal n, swapped
        if isinstance(w, dict) and set(w) == {"hits"}:
            hashable = lambda x: tuple(x) if isinstance(x, list) else x
            swapped += compare_lists([[hashable(i), s] for i, s in g["hits"]], [[hashable(i), s] for i, s in w["hits"]], where)
            n += 1
        elif isinstance(w, dict):
            assert isinstance(g, dict) and set(g) == set(w), where
            for key in w:
                walk(g[key], w[key], f"{where}.{key}")
        else:
            assert isinstance(w, list) and len(g) == len(w), where
            for i, (a, b) in enumerate(zip(g, w)):
                walk(a, b, f"{where}[{i}]")

    walk(got, want, "golden")
    return n, swapped
