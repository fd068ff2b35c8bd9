"""TEST INFRASTRUCTURE ONLY -- CPU restatement of what the two providers do with VectorBase hits to obtain message
ordinals (the step right after the hot path).  Pinned against the verbatim
`SqliteMessageTextIndex` (executed from /root/reference by oracle/ref_wrappers.py) in tests/test_reference_consumers.py.
"""

from __future__ import annotations

from typing import Callable, Sequence


def sqlite_messages_from_hits(
    hits: Sequence[tuple[int, float]],
    position_to_msg: dict[int, int] | Sequence[int],
    predicate: Callable[[int], bool] | None = None,
    max_matches: int | None = None,
) -> list[tuple[int, float]]:
    """storage/sqlite/messageindex.py:182-257: hits (index_position, score), best first ->
    rows looked up by `index_position IN (...)` (:193-210; positions without a row are dropped, :215) -> predicate on
    msg_id (:218-219) -> dict of best score per message in order of first appearance (:236-244) -> stable sort by score
    descending (:251) -> `[:max_matches]` when given (:254-255)."""
    scores: dict[int, float] = {}
    for pos, score in hits:
        if isinstance(position_to_msg, dict):
            if pos not in position_to_msg:
                continue
            msg = position_to_msg[pos]
        else:
            if not (0 <= pos < len(position_to_msg)) or position_to_msg[pos] < 0:
                continue
            msg = int(position_to_msg[pos])
        if predicate is not None and not predicate(msg):
            continue
        if msg not in scores:
            scores[msg] = score
        else:
            scores[msg] = max(scores[msg], score)
    out = sorted(scores.items(), key=lambda t: t[1], reverse=True)  # list.sort is stable
    return out if max_matches is None else out[:max_matches]


def sqlite_lookup_by_embedding(lookup, embedding, position_to_msg, max_matches=None, threshold_score=None, ordinals_to_search=None):
    """storage/sqlite/messageindex.py:296-326 `lookup_by_embedding` / `lookup_in_subset_by_embedding`:
    `lookup(embedding, max_hits, min_score)` is VectorBase.fuzzy_lookup_embedding -> [(position, score)]."""
    hits = lookup(embedding, max_matches, threshold_score)
    pred = None
    if ordinals_to_search is not None:
        members = set(ordinals_to_search)
        pred = members.__contains__
    return sqlite_messages_from_hits(hits, position_to_msg, pred, max_matches)


def memory_messages_from_hits(hits: Sequence[tuple[int, float]], position_to_msg: Sequence[int]) -> list[tuple[int, float]]:
    """storage/memory/messageindex.py:185-207 `to_scored_message_ordinals` over the locations of the hits
    (knowpro/textlocindex.py:179-183): best score per message, sorted by score descending (stable), no cut."""
    best: dict[int, float] = {}
    for pos, score in hits:
        msg = int(position_to_msg[pos])
        if msg not in best:
            best[msg] = score
        else:
            best[msg] = max(score, best[msg])
    return sorted(best.items(), key=lambda t: t[1], reverse=True)


def memory_lookup_in_subset_by_embedding(lookup_in_subset, embedding, position_to_msg, ordinals_to_search, max_matches=None, threshold_score=None):
    """storage/memory/messageindex.py:173-183 `lookup_in_subset_by_embedding`: the caller's MESSAGE ordinals are handed to
    `TextToTextLocationIndex.lookup_in_subset_by_embedding` (knowpro/textlocindex.py:164-177), which passes them on as ROW ordinals of
    the VectorBase (`fuzzy_lookup_embedding_in_subset`, vectorbase.py:203-230) and maps each hit row to its text location -- so with
    several chunks per message the rows searched are rows[ordinal], not the chunks of message `ordinal` -- then
    `to_scored_message_ordinals` (:185-207).  `lookup_in_subset(embedding, rows, max_hits, min_score)` -> [(row, score)]."""
    hits = lookup_in_subset(embedding, list(ordinals_to_search), max_matches, threshold_score)
    return memory_messages_from_hits(hits, position_to_msg)
