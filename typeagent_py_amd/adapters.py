"""Callers on either side of the VectorBase hot path (the "next" rows),
restated so that they use the device path in one submission instead of a Python loop.

Reference lines (/root/reference):
  * `TermEmbeddingIndex.lookup_terms`            storage/memory/reltermsindex.py:320-332
  * `SqliteRelatedTermsFuzzy.lookup_terms`       storage/sqlite/reltermsindex.py:259-271
      ("TODO: Some kind of batching?") -- both are `[await fuzzy_lookup(text, ...) for text in texts]`
  * message-ordinal aggregation                  storage/sqlite/messageindex.py:228-257,
                                                 storage/memory/messageindex.py:185-207
      (best score per message over its chunks, sorted by score, cut at max_matches)
  * raw little-endian float32 embedding files    knowpro/serialization.py:84-98, 114-136
      (`<name>_embeddings.bin`: related-term rows first, then message rows)
  * float32 BLOB columns of the SQLite provider  storage/sqlite/schema.py:71-81, 131-136, 193-197;
      reload loops storage/sqlite/messageindex.py:33-45, storage/sqlite/reltermsindex.py:144-156
"""

from __future__ import annotations

import os
from collections.abc import Callable, Sequence

import numpy as np

from .vectorbase import ScoredInt, VectorBase


async def lookup_texts_batched(
    vector_base: VectorBase,
    texts: Sequence[str],
    max_hits: int | None = None,
    min_score: float | None = None,
) -> list[list[ScoredInt]]:
    """== [await vector_base.fuzzy_lookup(t, max_hits, min_score) for t in texts], as ONE device
    submission: the texts are embedded in one (cache-aware) call and looked up by
    `fuzzy_lookup_embeddings`.  Defaults resolve like `fuzzy_lookup` (vectorbase.py:239-242):
    max_hits <- settings.max_matches (None -> 10), min_score <- settings.min_score."""
    texts = list(texts)
    if not texts:
        return []
    if max_hits is None:
        max_hits = vector_base.settings.max_matches
    if min_score is None:
        min_score = vector_base.settings.min_score
    embeddings = await vector_base.get_embeddings(texts)
    return vector_base.fuzzy_lookup_embeddings(np.asarray(embeddings, dtype=np.float32), max_hits=max_hits, min_score=min_score)


_patched: list[tuple[type, str, object]] = []


def install_batched_lookup_terms() -> dict:
    """When typeagent's related-terms indexes are importable, replace their two sequential `lookup_terms` bodies
    (storage/memory/reltermsindex.py:320-332, storage/sqlite/reltermsindex.py:259-271) by the batched form above.
    Returns {"patched": [class names], "skipped": {module name: reason}} -- a module that cannot be imported is reported,
    never silently ignored; any other failure propagates.  `uninstall_batched_lookup_terms()` restores the originals."""
    report: dict = {"patched": [], "skipped": {}}
    try:
        from typeagent.storage.memory import reltermsindex as mem  # type: ignore
    except ImportError as exc:
        report["skipped"]["typeagent.storage.memory.reltermsindex"] = f"{type(exc).__name__}: {exc}"
    else:
        async def lookup_terms(self, texts, max_hits=None, min_score=None):
            matches = await lookup_texts_batched(self._vectorbase, texts, max_hits, min_score)
            return [self.matches_to_terms(m) for m in matches]

        _patched.append((mem.TermEmbeddingIndex, "lookup_terms", mem.TermEmbeddingIndex.lookup_terms))
        mem.TermEmbeddingIndex.lookup_terms = lookup_terms
        report["patched"].append("typeagent.storage.memory.reltermsindex.TermEmbeddingIndex")
    try:
        from typeagent.knowpro import interfaces  # type: ignore
        from typeagent.storage.sqlite import reltermsindex as sql  # type: ignore
    except ImportError as exc:
        report["skipped"]["typeagent.storage.sqlite.reltermsindex"] = f"{type(exc).__name__}: {exc}"
    else:
        async def lookup_terms_sql(self, texts, max_hits=None, min_score=None):
            # == [await self.lookup_term(t, max_hits, min_score) for t in texts] (:259-271 over :158-179)
            matches = await lookup_texts_batched(self._vector_base, texts, max_hits, min_score)
            return [[interfaces.Term(self._terms_list[m.item], m.score) for m in ms if m.item < len(self._terms_list)] for ms in matches]

        _patched.append((sql.SqliteRelatedTermsFuzzy, "lookup_terms", sql.SqliteRelatedTermsFuzzy.lookup_terms))
        sql.SqliteRelatedTermsFuzzy.lookup_terms = lookup_terms_sql
        report["patched"].append("typeagent.storage.sqlite.reltermsindex.SqliteRelatedTermsFuzzy")
    return report


def uninstall_batched_lookup_terms() -> None:
    while _patched:
        cls, name, original = _patched.pop()
        setattr(cls, name, original)


def best_score_per_message(
    hits: Sequence[ScoredInt],
    row_to_message: Callable[[int], int] | Sequence[int] | np.ndarray,
    max_matches: int | None = None,
    accept: Callable[[int], bool] | None = None,
) -> list[ScoredInt]:
    """Chunk-row hits -> message hits: best score per message, sorted by score (stable), cut at
    `max_matches` -- the aggregation both providers apply after the VectorBase call
    (sqlite/messageindex.py:228-257; memory/messageindex.py:185-207).  `accept(message_ordinal)` is the
    subset filter the sqlite provider applies after the full scan (sqlite/messageindex.py:312-326)."""
    best: dict[int, float] = {}
    for h in hits:
        msg = int(row_to_message(h.item)) if callable(row_to_message) else int(row_to_message[h.item])
        if accept is not None and not accept(msg):
            continue
        if msg not in best or h.score > best[msg]:
            best[msg] = h.score
    out = [ScoredInt(m, s) for m, s in best.items()]
    out.sort(key=lambda x: x.score, reverse=True)
    return out if max_matches is None else out[:max_matches]


def _ensure_row_messages(vector_base: VectorBase, row_to_message) -> None:
    """Hand `row_to_message` to the index unless it already holds a snapshot of this very object that still covers it: the index
    keeps a COPY (a list the caller keeps appending to, or an array edited in place, must not alias the device map), so the
    snapshot is retaken when the object is another one, when its length changed, or when the index has outgrown it.  A caller that
    rewrites entries of a same-length map in place calls `vector_base.set_row_messages(map)` itself afterwards."""
    same = getattr(vector_base, "_row_messages_src", None) is row_to_message
    held = getattr(vector_base, "_row_messages", None)
    if not same or held is None or len(held) != len(row_to_message) or len(held) < len(vector_base):
        vector_base.set_row_messages(row_to_message)


def lookup_messages_by_embedding(
    vector_base: VectorBase,
    embedding,
    row_to_message,
    max_matches: int | None = None,
    threshold_score: float | None = None,
    accept: Callable[[int], bool] | Sequence[int] | set | None = None,
) -> list[ScoredInt]:
    """`SqliteMessageTextIndex.lookup_by_embedding` / `lookup_in_subset_by_embedding` restated on the device
    path (storage/sqlite/messageindex.py:296-326): ONE full-corpus top-`max_matches` chunk lookup, THEN the
    message-ordinal filter, THEN best score per message, THEN the cut -- in that order, so that it returns
    exactly what the sqlite provider returns (possibly fewer than `max_matches` messages).
    `accept`: the provider's `ordinals_set` as a collection of message ordinals -> the whole thing is one device
    submission (`VectorBase.lookup_messages_by_embedding`: lookup + bitmap filter + per-message reduction kernels); an
    arbitrary callable -> lookup on the device, aggregation on the host."""
    if accept is None or not callable(accept):
        if not callable(row_to_message):
            _ensure_row_messages(vector_base, row_to_message)
            return vector_base.lookup_messages_by_embedding(embedding, max_matches, threshold_score, accept_ordinals=accept)
        if accept is not None:
            members = set(int(x) for x in accept)
            accept = members.__contains__
    hits = vector_base.fuzzy_lookup_embedding(embedding, max_hits=max_matches, min_score=threshold_score)
    return best_score_per_message(hits, row_to_message, max_matches, accept)


def lookup_messages_in_subset(
    vector_base: VectorBase,
    embedding,
    rows_of_subset: Sequence[int],
    row_to_message,
    max_matches: int | None = None,
    threshold_score: float | None = None,
) -> list[ScoredInt]:
    """The memory provider's form (storage/memory/messageindex.py:173-207 via knowpro/textlocindex.py:164-177):
    a true subset gather on the device, then best score per message and the cut (one device submission when
    `row_to_message` is an array)."""
    if not callable(row_to_message):
        _ensure_row_messages(vector_base, row_to_message)
        out = vector_base.lookup_messages_in_subset_by_embedding(embedding, list(rows_of_subset), max_matches, threshold_score)
        return out if max_matches is None else out[:max_matches]
    hits = vector_base.fuzzy_lookup_embedding_in_subset(embedding, list(rows_of_subset), max_hits=max_matches, min_score=threshold_score)
    return best_score_per_message(hits, row_to_message, max_matches)


def load_sqlite_embeddings(
    db,
    vector_base: VectorBase,
    table: str = "MessageTextIndex",
    column: str = "embedding",
    order_by: str | None = None,
    key_column: str | None = None,
    fetch_rows: int = 8192,
) -> list:
    """Reload a VectorBase from the float32 BLOB column of a typeagent SQLite database the way the two
    sqlite indexes do at open time, without building one Python list of per-row arrays:
      * `SqliteMessageTextIndex.__init__`  -- `SELECT embedding FROM MessageTextIndex`
        (storage/sqlite/messageindex.py:33-45; schema.py:71-81)
      * `SqliteRelatedTermsFuzzy.__init__` -- `SELECT term, term_embedding FROM RelatedTermsFuzzy ORDER BY term`
        (storage/sqlite/reltermsindex.py:144-156; schema.py:131-136): pass table="RelatedTermsFuzzy",
        column="term_embedding", order_by="term", key_column="term".
    BLOBs are `ndarray.tobytes()` of float32 rows (schema.py:193-197).  Rows are fetched `fetch_rows` at a time,
    viewed with np.frombuffer and appended in bulk.  Returns the list of key_column values (row order = ordinals).
    `db` is a sqlite3.Connection."""
    for ident in (table, column, order_by, key_column):
        if ident is not None and not ident.replace("_", "").isalnum():
            raise ValueError(f"bad SQL identifier: {ident!r}")
    cols = f"{key_column}, {column}" if key_column else column
    sql = f"SELECT {cols} FROM {table}" + (f" ORDER BY {order_by}" if order_by else "")
    cur = db.cursor()
    cur.execute(sql)
    keys: list = []
    while True:
        rows = cur.fetchmany(fetch_rows)
        if not rows:
            break
        blobs = [r[-1] for r in rows]
        if any(b is None for b in blobs):
            raise ValueError(f"NULL embedding in {table}.{column}")
        width = len(blobs[0]) // 4
        if width == 0 or any(len(b) != width * 4 for b in blobs):
            raise ValueError("embedding BLOBs of unequal size")
        block = np.frombuffer(b"".join(blobs), dtype="<f4").reshape(len(blobs), width)
        vector_base.add_embeddings(None, block)  # size mismatch vs the index -> ValueError, as in the reference
        if key_column:
            keys.extend(r[0] for r in rows)
    return keys


def load_embeddings_bin(
    path: str,
    embedding_size: int,
    related_count: int,
    message_count: int,
    related_vb: VectorBase | None = None,
    message_vb: VectorBase | None = None,
    chunk_rows: int = 1 << 18,
):
    """Stream a `<name>_embeddings.bin` sidecar (raw little-endian float32 rows: `related_count`
    related-term rows, then `message_count` message rows; counts and width come from the
    `embeddingFileHeader` of `<name>_data.json`, knowpro/serialization.py:84-98, 207-221) straight into
    the given VectorBases in `chunk_rows`-row pieces (np.memmap -> add_embeddings), so that no second
    full copy of the file is built on the host.  Returns (related_rows, message_rows) loaded."""
    expected = (related_count + message_count) * embedding_size * 4
    actual = os.path.getsize(path)
    if actual != expected:
        raise ValueError(f"{path}: {actual} bytes, expected {expected} for {related_count}+{message_count} rows of {embedding_size} float32")
    rows = np.memmap(path, dtype="<f4", mode="r", shape=(related_count + message_count, embedding_size))
    done = [0, 0]
    for which, (vb, lo, hi) in enumerate(((related_vb, 0, related_count), (message_vb, related_count, related_count + message_count))):
        if vb is None:
            continue
        for a in range(lo, hi, chunk_rows):
            b = min(hi, a + chunk_rows)
            vb.add_embeddings(None, np.ascontiguousarray(rows[a:b], dtype=np.float32))
            done[which] += b - a
    return tuple(done)
