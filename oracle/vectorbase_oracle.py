"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the VectorBase kNN hot path.

A numpy restatement of the arithmetic of the reference's
`src/typeagent/aitools/vectorbase.py` (all file:line citations below are
relative to /root/reference), written as free functions over plain arrays so it
can run on the GPU box where /root/reference does not exist.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker / the reported baseline.  The
product package (`typeagent_py_amd`) never imports it and has no CPU fallback.

Pinning: `tests/test_oracle_vs_reference.py` checks every function here against
the *verbatim* reference class (loaded by `oracle/ref_loader.py`) when
/root/reference is present, and `tests/test_oracle_golden.py` checks it against
the committed golden vectors in `tests/golden/` (generated from the verbatim
reference by `tests/golden/make_golden.py`) everywhere else.  The reference's
own known-answer tests (tests/test_vectorbase.py:239-252,
tests/test_benchmark_embeddings.py:229-277) are part of those goldens.

Where the arithmetic really lives: numpy + OpenBLAS `sgemv` (third-party,
numpy>=2.2.6 in pyproject.toml:35).  Summation order inside sgemv and the order
of equal keys inside argpartition/argsort are unspecified by the reference, so
`check_topk_parity` below compares modulo fp32 (near-)ties -- see its docstring.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, Sequence

import numpy as np

DEFAULT_MAX_HITS = 10  # vectorbase.py:170-171, 210-211
DEFAULT_MIN_SCORE_ARG = 0.0  # vectorbase.py:172-173, 212-213


# --------------------------------------------------------------------------
# elementwise pieces
# --------------------------------------------------------------------------
def cosine_to_score(cosine: np.ndarray) -> np.ndarray:
    """vectorbase.py:44-47 -- clip((c + 1) / 2, 0, 1); stays float32 for f32 input."""
    shifted = cosine + 1.0
    halved = shifted / 2.0
    return np.clip(halved, 0.0, 1.0)


def l2_normalize_rows(x: np.ndarray) -> np.ndarray:
    """model_adapters.py:181-183 -- rows / ||row||_2, zero-norm rows left as they are."""
    x = np.asarray(x, dtype=np.float32)
    norms = np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    norms = np.where(norms > 0, norms, np.float32(1.0))
    return (x / norms).astype(np.float32)


def scores_full(vectors: np.ndarray, query: np.ndarray) -> np.ndarray:
    """vectorbase.py:176 -- the whole score vector for one query (float32[N] for f32 inputs)."""
    return cosine_to_score(np.dot(vectors, query))


# --------------------------------------------------------------------------
# selection
# --------------------------------------------------------------------------
def _select_desc(scores: np.ndarray, max_hits: int, min_score) -> tuple[np.ndarray, np.ndarray]:
    """vectorbase.py:179-190 / 219-230: threshold, then top-`max_hits` in descending order.

    Returns (positions into `scores`, their scores).  `min_score` is compared
    the way numpy does it for a Python scalar against a float32 array (NEP 50
    weak scalar => as float32); pass a numpy float64 scalar to get a float64
    compare, exactly like the reference would.
    """
    passing = np.flatnonzero(scores >= min_score)
    if passing.size == 0:
        return passing, scores[:0]
    kept = scores[passing]
    if passing.size <= max_hits:
        order = np.argsort(kept)[::-1]
    else:
        # NB: max_hits == 0 makes both slices `[-0:]` == everything (quirk 6 in SURVEY appendix A)
        head = np.argpartition(kept, -max_hits)[-max_hits:]
        order = head[np.argsort(kept[head])[::-1]]
    return passing[order], kept[order]


def lookup(
    vectors: np.ndarray,
    query: np.ndarray,
    max_hits: int | None = None,
    min_score: float | None = None,
    predicate: Callable[[int], bool] | None = None,
) -> list[tuple[int, float]]:
    """vectorbase.py:163-201 `fuzzy_lookup_embedding` -> [(ordinal, score)] best first."""
    if max_hits is None:
        max_hits = DEFAULT_MAX_HITS
    if min_score is None:
        min_score = DEFAULT_MIN_SCORE_ARG
    if len(vectors) == 0:  # :174-175
        return []
    scores = scores_full(vectors, query)
    if predicate is None:
        pos, sc = _select_desc(scores, max_hits, min_score)
        return [(int(p), float(s)) for p, s in zip(pos, sc)]
    # :191-201 predicate path: python filter over every survivor, stable sort, cut
    survivors = np.flatnonzero(scores >= min_score)
    kept = [(int(i), float(scores[i])) for i in survivors if predicate(int(i))]
    kept.sort(key=lambda t: t[1], reverse=True)  # stable => ties in ascending ordinal
    return kept[:max_hits]


def lookup_in_subset(
    vectors: np.ndarray,
    query: np.ndarray,
    ordinals_of_subset: Sequence[int],
    max_hits: int | None = None,
    min_score: float | None = None,
) -> list[tuple[int, float]]:
    """vectorbase.py:203-230 `fuzzy_lookup_embedding_in_subset`.

    Duplicates in the subset give duplicate hits; a negative ordinal reads the
    wrapped row but is reported as given; out of range raises IndexError
    (numpy fancy indexing, :218).
    """
    if max_hits is None:
        max_hits = DEFAULT_MAX_HITS
    if min_score is None:
        min_score = DEFAULT_MIN_SCORE_ARG
    if len(ordinals_of_subset) == 0 or len(vectors) == 0:  # :214-215
        return []
    subset = np.asarray(ordinals_of_subset)
    scores = cosine_to_score(np.dot(vectors[subset], query))
    pos, sc = _select_desc(scores, max_hits, min_score)
    return [(int(subset[p]), float(s)) for p, s in zip(pos, sc)]


def lookup_batch(
    vectors: np.ndarray,
    queries: np.ndarray,
    max_hits: int | None = None,
    min_score: float | None = None,
) -> list[list[tuple[int, float]]]:
    """The reference has no batch entry point; its semantics for many queries are
    sequential calls (storage/memory/reltermsindex.py:320-332, "TODO: Some kind of
    batching?" at storage/sqlite/reltermsindex.py:259-271)."""
    return [lookup(vectors, q, max_hits, min_score) for q in queries]


def lookup_chunked(
    chunks: Iterable[np.ndarray],
    query: np.ndarray,
    max_hits: int,
    min_score: float = 0.0,
) -> list[tuple[int, float]]:
    """Oracle for corpora larger than host RAM: run `lookup` per row-chunk (each
    an independent reference-sized VectorBase), offset the ordinals, keep the
    best `max_hits` overall.  Exact because a row's score depends only on that
    row and the query (vectorbase.py:176)."""
    best: list[tuple[int, float]] = []
    base = 0
    for chunk in chunks:
        part = lookup(chunk, query, max_hits, min_score)
        best.extend((base + i, s) for i, s in part)
        base += len(chunk)
    best.sort(key=lambda t: (-t[1], t[0]))
    return best[:max_hits]


# --------------------------------------------------------------------------
# parity checking modulo fp32 near-ties
# --------------------------------------------------------------------------
SCORE_TOL = 1e-5  # BASELINE.json north_star: cosine scores within 1e-5 (fp32)
# Scores in [0.5, 1] have an fp32 spacing of 2^-24 ~ 6e-8; a different summation
# order moves the cosine by ~4.5e-8 at D=1536.  Two rows
# whose reference scores are closer than this are a "near tie": their relative
# order is not defined by the reference (it depends on sgemv's summation order).
TIE_EPS = 4 * 2.0**-24


def tie_eps_at(score: float) -> float:
    """Near-tie width at a given score.  The reference's own float32 rounding noise grows with the size of the dot product
    (partial sums of magnitude |cos| are rounded to 2^-24 |cos| each): measured on this host (numpy/OpenBLAS sgemv against a
    float64 dot, D = 1536, the best 64 rows of 64 queries), |score_f32 - score_f64| reaches 4.5e-8 on isotropic data (top scores
    ~0.57, |cos| ~0.14) and 1.6e-7 on the clustered corpus (scores ~0.999, |cos| ~1) -- tools/oracle_noise.py.  TIE_EPS = 4 * 2^-24
    is ~5x the former; the same factor at |cos| = 1 is 16 * 2^-24 = 9.5e-7, still 10x inside the 1e-5 score tolerance.  TIE_EPS
    up to |cos| = |2 score - 1| = 0.25 (every isotropic workload: unchanged policy), linear from there to |cos| = 1."""
    if score != score:
        return TIE_EPS
    c = min(1.0, abs(2.0 * float(score) - 1.0))
    return (4.0 + 12.0 * max(0.0, c - 0.25) / 0.75) * 2.0**-24


@dataclass
class ParityReport:
    k_returned: int
    exact_positions: int
    tie_permuted_positions: int
    threshold_ambiguous: int

    @property
    def ordinals_bit_exact(self) -> bool:
        return self.tie_permuted_positions == 0


def check_topk_parity(
    ref_scores: np.ndarray,
    got_items: Sequence[int],
    got_scores: Sequence[float],
    max_hits: int,
    min_score: float = 0.0,
    score_tol: float = SCORE_TOL,
    tie_eps: float | None = None,
    candidate_ordinals: np.ndarray | None = None,
) -> ParityReport:
    """Assert that (got_items, got_scores) is the reference's answer for the
    score vector `ref_scores` (= `scores_full(V, q)`), modulo near-ties.

    Rules (SURVEY.md section 7 "Exact-ordinal parity under ties/near-ties"):
      1. every returned score is within `score_tol` of the reference score of that row;
      2. results are in descending score order;
      3. the returned ordinal *sequence* equals the reference's wherever the
         reference scores involved are separated by more than the near-tie width (`tie_eps`, default
         `tie_eps_at(score)`: 4 * 2^-24 around score 0.5, growing with |cos| to 16 * 2^-24 at score 1);
         rows inside a near-tie group (including a group straddling rank k, or
         straddling `min_score`) may be permuted / swapped;
      4. the count is min(max_hits, #survivors) up to threshold-ambiguous rows.

    `candidate_ordinals`: for subset searches, ref_scores[i] belongs to ordinal
    candidate_ordinals[i] (duplicates allowed); default arange(N).
    """
    ref_scores = np.asarray(ref_scores, dtype=np.float32)
    n = ref_scores.shape[0]
    got_items = [int(i) for i in got_items]
    got_scores = np.asarray(got_scores, dtype=np.float64)
    thr32 = float(np.float32(min_score)) if not isinstance(min_score, np.floating) else float(min_score)
    eps_of = (lambda sc: tie_eps) if tie_eps is not None else tie_eps_at
    thr_eps = eps_of(thr32)
    finite = ~np.isnan(ref_scores)
    sure = finite & (ref_scores >= thr32 + thr_eps)
    maybe = finite & (ref_scores >= thr32 - thr_eps) & ~sure
    n_sure, n_maybe = int(sure.sum()), int(maybe.sum())
    kcap = max_hits if max_hits > 0 else n  # max_hits==0 quirk: everything
    lo, hi = min(kcap, n_sure), min(kcap, n_sure + n_maybe)
    assert lo <= len(got_items) <= hi, f"count {len(got_items)} not in [{lo},{hi}]"
    assert len(got_scores) == len(got_items)

    # map ordinal -> candidate positions
    if candidate_ordinals is None:
        def ref_of(item: int, used: set) -> float:
            assert 0 <= item < n, f"ordinal {item} out of range"
            assert item not in used, f"ordinal {item} returned twice"
            used.add(item)
            return float(ref_scores[item])
    else:
        cand = np.asarray(candidate_ordinals)
        slots: dict[int, list[int]] = {}
        for pos, o in enumerate(cand.tolist()):
            slots.setdefault(int(o), []).append(pos)

        def ref_of(item: int, used: set) -> float:
            assert item in slots and slots[item], f"ordinal {item} not in subset (or returned too often)"
            pos = slots[item].pop(0)
            used.add(pos)
            return float(ref_scores[pos])

    used: set = set()
    ref_for_got = np.array([ref_of(it, used) for it in got_items], dtype=np.float64)
    # rule 1
    if len(got_items):
        err = np.abs(ref_for_got - got_scores)
        assert float(err.max()) <= score_tol, f"score error {err.max():.3e} > {score_tol}"
        # rule 2
        assert np.all(np.diff(got_scores) <= 0), "returned scores not descending"
        assert np.all(ref_for_got >= thr32 - thr_eps), "returned a row below min_score"

    # reference ranking (score desc, position asc) over survivors
    elig = np.flatnonzero(sure | maybe)
    order = elig[np.lexsort((elig, -ref_scores[elig].astype(np.float64)))]
    ref_sorted = ref_scores[order].astype(np.float64)
    exact = permuted = 0
    k = len(got_items)
    if k:
        # rule 3: position i must hold a row whose reference score is within
        # tie_eps of the reference's i-th best score
        for i in range(k):
            want = ref_sorted[i]
            have = ref_for_got[i]
            assert abs(want - have) <= eps_of(want), (
                f"rank {i}: got ordinal {got_items[i]} (ref score {have:.9f}) but reference rank-{i} "
                f"score is {want:.9f}"
            )
            ref_item = int(order[i]) if candidate_ordinals is None else int(np.asarray(candidate_ordinals)[order[i]])
            if ref_item == got_items[i]:
                exact += 1
            else:
                permuted += 1
        # nothing clearly better was left out
        if k < len(order):
            worst = ref_for_got.min()
            mask = np.ones(n, dtype=bool)
            if candidate_ordinals is None:
                mask[got_items] = False
            else:
                mask[list(used)] = False
            rest = ref_scores[mask & (sure | maybe)]
            if rest.size and k >= kcap:
                assert float(rest.max()) <= worst + eps_of(worst), (
                    f"omitted a row with ref score {rest.max():.9f} > worst returned {worst:.9f}"
                )
    return ParityReport(k, exact, permuted, n_maybe)


def f32_threshold(min_score) -> np.float32:
    """The float32 threshold `t` such that, for float32 `s`, `s >= t` is what numpy
    evaluates for `s >= min_score` (vectorbase.py:179): a Python float/int is a
    weak scalar and is cast to float32 (NEP 50; f32(0.85) = 0.8500000238...); a
    numpy float64 scalar forces a float64 compare, equivalent to the smallest
    float32 >= it."""
    if isinstance(min_score, np.floating) and not isinstance(min_score, np.float32):
        d = float(min_score)
        if d != d:
            return np.float32(np.nan)
        t = np.float32(d)
        if float(t) < d:
            t = np.nextafter(t, np.float32(np.inf), dtype=np.float32)
        return t
    with np.errstate(over="ignore"):
        return np.float32(min_score)


# --------------------------------------------------------------------------
# corpora larger than one reference-sized VectorBase (bench.py / full-size GPU tests)
# --------------------------------------------------------------------------
def scores_full_chunked(chunks: Iterable[np.ndarray], queries: np.ndarray) -> np.ndarray:
    """float32 [nq, N] score matrix for a handful of queries over a corpus delivered in row chunks (each chunk a
    reference-sized float32 matrix).  Per chunk this is vectorbase.py:176 for every query at once (`np.dot(chunk,
    Q.T)`: sgemm instead of nq sgemv calls -- the summation order differs from the reference's by the usual fp32
    noise, which `check_topk_parity` already treats as a near-tie matter)."""
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    parts = [cosine_to_score(np.dot(chunk, queries.T)).T for chunk in chunks]
    return np.ascontiguousarray(np.concatenate(parts, axis=1), dtype=np.float32)


def check_topk_parity_large(
    ref_scores: np.ndarray,
    got_items: Sequence[int],
    got_scores: Sequence[float],
    max_hits: int,
    min_score: float = 0.0,
    margin: int = 256,
) -> tuple[ParityReport, int]:
    """`check_topk_parity` for multi-million-row score vectors: the reference ranking is only needed down to rank
    `max_hits` (+ `margin` rows of slack for near-tie groups), so the check runs on the best `max_hits + margin`
    reference rows.  A returned ordinal outside that set fails the check, as it should.  Also returns the number of
    adjacent reference pairs within the near-tie width among ranks 0..max_hits (how many near-ties the answer contains)."""
    ref_scores = np.asarray(ref_scores, dtype=np.float32)
    n = ref_scores.shape[0]
    keep = min(n, max_hits + margin)
    clean = np.where(np.isnan(ref_scores), np.float32(-1.0), ref_scores)
    top = np.argpartition(-clean, keep - 1)[:keep] if keep < n else np.arange(n)
    top = top[np.lexsort((top, -clean[top].astype(np.float64)))]
    rep = check_topk_parity(ref_scores[top], got_items, got_scores, max_hits, min_score, candidate_ordinals=top)
    head = clean[top[: max_hits + 1]].astype(np.float64)
    near = int(sum(abs(a - b) <= tie_eps_at(a) for a, b in zip(head[:-1], head[1:])))
    return rep, near
