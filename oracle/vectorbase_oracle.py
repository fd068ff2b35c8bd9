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
# parity checking modulo fp32 near-ties, with a float64 referee
# --------------------------------------------------------------------------
SCORE_TOL = 1e-5  # BASELINE.json north_star: cosine scores within 1e-5 (fp32)
# Scores in [0.5, 1] have an fp32 spacing of 2^-24 ~ 6e-8; a different summation
# order moves the cosine by ~4.5e-8 at D=1536 on isotropic data.
# Two rows whose reference scores are closer than this are a "near tie": their relative
# order is not defined by the reference (it depends on sgemv's summation order).
# This constant is the WHOLE rule when no referee is supplied (small isotropic cases).
TIE_EPS = 4 * 2.0**-24


def scores_f64(rows: np.ndarray, query: np.ndarray) -> np.ndarray:
    """The exact score of vectorbase.py:176 + :44-47 for the given float32 rows: the same formula evaluated in float64
    (the float32 inputs are exact in float64; a 1536-term float64 dot is good to ~1e-14)."""
    dots = np.asarray(rows, dtype=np.float64) @ np.asarray(query, dtype=np.float64)
    return np.clip((dots + 1.0) / 2.0, 0.0, 1.0)


def f64_referee(vectors: np.ndarray, query: np.ndarray) -> Callable[[np.ndarray], np.ndarray]:
    """Referee for `check_topk_parity`: positions (into the score vector handed to the checker) -> float64 scores.
    For a subset search pass `vectors[subset]`."""
    def referee(positions: np.ndarray) -> np.ndarray:
        return scores_f64(vectors[np.asarray(positions, dtype=np.int64)], query)
    referee.dim = int(np.asarray(query).shape[-1])  # lets the checker price legitimate float32 accumulation noise
    return referee


def fp32_accumulation_scale(dim: int | None) -> float:
    """The random-walk scale, in score units, of a float32 dot product of `dim` terms whose partial sums stay within [-1, 1]: every one
    of the `dim` additions rounds by at most 2^-24 -> sqrt(dim) * 2^-24 on the cosine, half of that on the score (vectorbase.py:44-47
    halves it).  1.2e-6 at dim = 1536; the worst case (dim * 2^-24) is 40x that.  Measured on the device: 3e-7 for the MFMA tiles at
    |cos| ~ 1 (96 dependent accumulations of 16-term blocks), 6e-8 for the streaming kernels (24 terms per lane, then a tree)."""
    return 0.0 if not dim else 0.5 * float(np.sqrt(dim)) * 2.0**-24


@dataclass
class ParityReport:
    k_returned: int
    exact_positions: int
    tie_permuted_positions: int
    threshold_ambiguous: int
    # ---- filled in when a float64 referee was supplied (else None / 0)
    refereed: bool = False
    noise_ref: float | None = None          # max |reference float32 score - float64 score| over the rows looked at
    noise_gpu: float | None = None          # max |returned score - float64 score| over the returned rows
    tie_width: float | None = None          # 2 * (noise_ref + noise_gpu): the widest float64 gap two rows swapped at one rank can have
    max_permuted_gap: float = 0.0           # widest float64 gap between the row returned and the reference's row at a permuted rank
    gpu_inversions_vs_f64: int = 0          # pairs the returned answer orders (or drops) against the float64 truth
    reference_inversions_vs_f64: int = 0    # the same count for the reference's float32 answer
    max_inverted_gap_gpu: float = 0.0       # widest float64 gap of such a pair
    max_inverted_gap_ref: float = 0.0

    @property
    def ordinals_bit_exact(self) -> bool:
        return self.tie_permuted_positions == 0


def _inversions(f64_listed: np.ndarray, f64_omitted: np.ndarray) -> tuple[int, float]:
    """Pairs that an answer gets wrong against the float64 truth: (i, j), i ahead of j in the list, truth[i] < truth[j]; and
    (listed r, omitted o) with truth[o] > truth[r].  Returns (count, widest truth gap of such a pair).  Exact float64 ties are
    not inversions."""
    a = np.asarray(f64_listed, dtype=np.float64)
    count, worst = 0, 0.0
    if a.size > 1:
        d = a[None, :] - a[:, None]          # d[i, j] = truth[j] - truth[i]
        bad = np.triu(d > 0.0, k=1)          # j behind i but truly better
        count += int(bad.sum())
        if bad.any():
            worst = max(worst, float(d[bad].max()))
    if a.size and f64_omitted.size:
        d = f64_omitted[None, :] - a[:, None]
        bad = d > 0.0
        count += int(bad.sum())
        if bad.any():
            worst = max(worst, float(d[bad].max()))
    return count, worst


def check_topk_parity(
    ref_scores: np.ndarray,
    got_items: Sequence[int],
    got_scores: Sequence[float],
    max_hits: int,
    min_score: float = 0.0,
    score_tol: float = SCORE_TOL,
    tie_eps: float | None = None,
    candidate_ordinals: np.ndarray | None = None,
    referee: Callable[[np.ndarray], np.ndarray] | None = None,
    referee_slack: int = 64,
) -> ParityReport:
    """Assert that (got_items, got_scores) is the reference's answer for the
    score vector `ref_scores` (= `scores_full(V, q)`), modulo near-ties.

    Rules (SURVEY.md section 7 "Exact-ordinal parity under ties/near-ties"):
      1. every returned score is within `score_tol` of the reference score of that row;
      2. results are in descending score order;
      3. the returned ordinal *sequence* equals the reference's wherever the reference itself is determinate: rows inside a
         near-tie group (including a group straddling rank k, or straddling `min_score`) may be permuted / swapped;
      4. the count is min(max_hits, #survivors) up to threshold-ambiguous rows.

    What a near tie is.  The reference's order among float32 scores closer than its own summation noise is undefined
    (vectorbase.py:176 is OpenBLAS sgemv, :183-187 numpy's introselect).
      * Without a `referee` the rule is the constant `TIE_EPS` = 4 * 2^-24 on the reference's float32 scores (`tie_eps` overrides).
      * With a `referee` (positions -> float64 scores: `f64_referee(V, q)`) nothing is hand-set.  The checker computes the float64
        truth T for the reference's best `max_hits + referee_slack` rows and for every returned row, MEASURES
            noise_ref = max |ref32 - T|,   noise_gpu = max |returned score - T|
        and accepts a different row at rank i only if its truth is within  tie_width = 2 * (noise_ref + noise_gpu)  of the truth of
        the reference's row at that rank.  (Order statistics are 1-Lipschitz: the i-th best of scores perturbed by <= e lies within e of
        the i-th best truth, and that row's own truth within another e -- so two correct rankings, one per arithmetic, can differ at a
        rank only by rows this close.  A row swapped over a wider gap is a wrong answer, whatever the noise.)
        The measured GPU noise itself must stay within max(TIE_EPS, 4 * noise_ref, fp32_accumulation_scale(dim)): the device may not be
        sloppier than a few times the reference's own arithmetic or than float32 accumulation of `dim` terms explains (OpenBLAS's
        blocked sgemv lands anywhere between 4e-8 and 1.6e-7 at |cos| ~ 1, query by query), so "measured" cannot excuse a defect.
        The report also counts, against the truth, the pairs each answer orders wrongly (`gpu_inversions_vs_f64`,
        `reference_inversions_vs_f64`): the returned answer is at least as right as OpenBLAS's when the former <= the latter.

    `candidate_ordinals`: for subset searches, ref_scores[i] belongs to ordinal
    candidate_ordinals[i] (duplicates allowed); default arange(N).
    """
    ref_scores = np.asarray(ref_scores, dtype=np.float32)
    n = ref_scores.shape[0]
    got_items = [int(i) for i in got_items]
    got_scores = np.asarray(got_scores, dtype=np.float64)
    thr32 = float(np.float32(min_score)) if not isinstance(min_score, np.floating) else float(min_score)
    base_eps = TIE_EPS if tie_eps is None else float(tie_eps)
    finite = ~np.isnan(ref_scores)

    # map ordinal -> candidate positions
    if candidate_ordinals is None:
        def pos_of(item: int, used: set) -> int:
            assert 0 <= item < n, f"ordinal {item} out of range"
            assert item not in used, f"ordinal {item} returned twice"
            used.add(item)
            return item
    else:
        cand = np.asarray(candidate_ordinals)
        slots: dict[int, list[int]] = {}
        for pos, o in enumerate(cand.tolist()):
            slots.setdefault(int(o), []).append(pos)

        def pos_of(item: int, used: set) -> int:
            assert item in slots and slots[item], f"ordinal {item} not in subset (or returned too often)"
            pos = slots[item].pop(0)
            used.add(pos)
            return pos

    used: set = set()
    got_pos = np.array([pos_of(it, used) for it in got_items], dtype=np.int64)
    ref_for_got = ref_scores[got_pos].astype(np.float64) if len(got_items) else np.zeros(0)
    assert len(got_scores) == len(got_items)
    # rule 1
    if len(got_items):
        err = np.abs(ref_for_got - got_scores)
        assert float(err.max()) <= score_tol, f"score error {err.max():.3e} > {score_tol}"
        # rule 2
        assert np.all(np.diff(got_scores) <= 0), "returned scores not descending"

    # ---- the referee: float64 truth for the rows that matter, measured noise of both arithmetics
    kcap = max_hits if max_hits > 0 else n  # max_hits==0 quirk: everything
    truth: dict[int, float] = {}
    noise_ref = noise_gpu = width = None
    width_eps = base_eps   # near-tie width on float64 gaps (referee) or on reference float32 scores (none)
    thr_eps = base_eps
    if referee is not None:
        prelim = np.flatnonzero(finite & (ref_scores >= thr32 - score_tol))
        head = min(prelim.size, min(kcap, max(len(got_items), 1)) + referee_slack)
        if head < prelim.size:
            top = prelim[np.argpartition(-ref_scores[prelim], head - 1)[:head]]
        else:
            top = prelim
        looked = np.unique(np.concatenate([top, got_pos])) if len(got_items) else np.unique(top)
        if looked.size:
            t = np.asarray(referee(looked), dtype=np.float64)
            truth = dict(zip(looked.tolist(), t.tolist()))
            noise_ref = float(np.max(np.abs(ref_scores[looked].astype(np.float64) - t)))
        else:
            noise_ref = 0.0
        if len(got_items):
            t_got = np.array([truth[int(p)] for p in got_pos])
            noise_gpu = float(np.max(np.abs(got_scores - t_got)))
        else:
            noise_gpu = 0.0
        noise_cap = max(TIE_EPS, 4.0 * noise_ref, fp32_accumulation_scale(getattr(referee, "dim", None)))
        assert noise_gpu <= noise_cap, (
            f"device scores are noisier than float32 arithmetic explains: max |score - float64| = {noise_gpu:.3e} against the reference's "
            f"{noise_ref:.3e} (cap {noise_cap:.3e})")
        width = 2.0 * (noise_ref + noise_gpu)
        width_eps = width
        thr_eps = max(base_eps, noise_ref + noise_gpu)

    sure = finite & (ref_scores >= thr32 + thr_eps)
    maybe = finite & (ref_scores >= thr32 - thr_eps) & ~sure
    n_sure, n_maybe = int(sure.sum()), int(maybe.sum())
    lo, hi = min(kcap, n_sure), min(kcap, n_sure + n_maybe)
    assert lo <= len(got_items) <= hi, f"count {len(got_items)} not in [{lo},{hi}]"
    if len(got_items):
        assert np.all(ref_for_got >= thr32 - thr_eps), "returned a row below min_score"

    def val(pos: int) -> float:  # what near-tie decisions are taken on: the truth when there is a referee
        return truth[int(pos)] if referee is not None else float(ref_scores[int(pos)])

    # reference ranking (score desc, position asc) over survivors
    elig = np.flatnonzero(sure | maybe)
    order = elig[np.lexsort((elig, -ref_scores[elig].astype(np.float64)))]
    exact = permuted = 0
    max_perm_gap = 0.0
    k = len(got_items)
    rep = ParityReport(k, 0, 0, n_maybe)
    if k:
        # rule 3: position i must hold the reference's row, or a row in a near tie with it
        for i in range(k):
            ref_pos = int(order[i])
            if referee is not None and ref_pos not in truth:   # (cannot happen with the default slack unless k rows tie: look it up)
                truth[ref_pos] = float(np.asarray(referee(np.array([ref_pos])), dtype=np.float64)[0])
            ref_item = ref_pos if candidate_ordinals is None else int(np.asarray(candidate_ordinals)[ref_pos])
            if ref_item == got_items[i]:
                exact += 1
                continue
            want, have = val(ref_pos), val(int(got_pos[i]))
            gap = abs(want - have)
            assert gap <= width_eps, (
                f"rank {i}: got ordinal {got_items[i]} ({'float64' if referee is not None else 'ref'} score {have:.9f}) but the reference's "
                f"rank-{i} row {ref_item} scores {want:.9f}: gap {gap:.3e} > near-tie width {width_eps:.3e}")
            permuted += 1
            max_perm_gap = max(max_perm_gap, gap)
        # nothing clearly better was left out
        if k < len(order) and k >= kcap:
            mask = np.zeros(n, dtype=bool)
            mask[elig] = True
            mask[list(used)] = False
            if referee is not None:
                rest = np.array([truth[p] for p in truth if mask[p]], dtype=np.float64)
                worst = min(truth[int(p)] for p in got_pos)
            else:
                rest = ref_scores[mask].astype(np.float64)
                worst = float(ref_for_got.min())
            if rest.size:
                assert float(rest.max()) <= worst + width_eps, (
                    f"omitted a row scoring {rest.max():.9f} > worst returned {worst:.9f} (near-tie width {width_eps:.3e})")
    rep.exact_positions, rep.tie_permuted_positions = exact, permuted
    if referee is not None:
        rep.refereed = True
        rep.noise_ref, rep.noise_gpu, rep.tie_width, rep.max_permuted_gap = noise_ref, noise_gpu, width, max_perm_gap
        # how right is each answer against the truth?  (listed rows, and the looked-at rows each answer left out)
        in_elig = np.zeros(n, dtype=bool)
        in_elig[elig] = True
        looked_elig = [p for p in truth if in_elig[p]]
        got_set = set(int(p) for p in got_pos)
        ref_list = [int(p) for p in order[: min(len(order), kcap, max(k, 1) if k else kcap)]] if k else []
        for p in ref_list:
            if p not in truth:
                truth[p] = float(np.asarray(referee(np.array([p])), dtype=np.float64)[0])
        ref_set = set(ref_list)
        full = k >= kcap  # an answer cut at max_hits "omits" rows; one that returns every survivor does not
        om_gpu = np.array([truth[p] for p in looked_elig if p not in got_set], dtype=np.float64) if full else np.zeros(0)
        om_ref = np.array([truth[p] for p in looked_elig if p not in ref_set], dtype=np.float64) if full else np.zeros(0)
        rep.gpu_inversions_vs_f64, rep.max_inverted_gap_gpu = _inversions(np.array([truth[int(p)] for p in got_pos]), om_gpu)
        rep.reference_inversions_vs_f64, rep.max_inverted_gap_ref = _inversions(np.array([truth[p] for p in ref_list]), om_ref)
    return rep


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
    referee: Callable[[np.ndarray], np.ndarray] | None = None,
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
    inner = None
    if referee is not None:
        def inner(positions):
            return referee(top[np.asarray(positions, dtype=np.int64)])
        inner.dim = getattr(referee, "dim", None)
    rep = check_topk_parity(ref_scores[top], got_items, got_scores, max_hits, min_score, candidate_ordinals=top, referee=inner)
    head = clean[top[: max_hits + 1]].astype(np.float64)
    near_width = rep.tie_width if rep.refereed else TIE_EPS
    near = int(sum(abs(a - b) <= near_width for a, b in zip(head[:-1], head[1:])))
    return rep, near


class ChunkedReferee:
    """float64 truth for corpora delivered in row chunks (bench.py, full-size GPU tests): while a chunk is in host memory the truth is
    computed for the rows that can matter -- the chunk's best `keep` rows per query by the float32 reference score, plus the ordinals the
    device returned -- and remembered per query.  The whole corpus' best `keep` rows are among the chunks' best `keep` rows."""

    def __init__(self, queries: np.ndarray, returned: Sequence[Sequence[int]], keep: int, restrict: np.ndarray | None = None):
        """restrict (optional): the ordinals of a SUBSET search (vectorbase.py:203-230) -- the rows that can matter are the best `keep` of the
        subset's rows in each chunk, not of the chunk."""
        self.queries = np.ascontiguousarray(queries, dtype=np.float32)
        self.returned = [np.unique(np.asarray(r, dtype=np.int64)) for r in returned]
        self.keep = int(keep)
        self.restrict = None if restrict is None else np.unique(np.asarray(restrict, dtype=np.int64))
        self.truth: list[dict[int, float]] = [dict() for _ in range(len(self.queries))]

    def see_chunk(self, base: int, chunk: np.ndarray, scores32: np.ndarray) -> None:
        """scores32: float32 [rows of the chunk, nq] reference scores of this chunk."""
        rows = chunk.shape[0]
        cand = None
        if self.restrict is not None:
            cand = self.restrict[(self.restrict >= base) & (self.restrict < base + rows)] - base
        for j in range(len(self.queries)):
            col = np.where(np.isnan(scores32[:, j]), np.float32(-1.0), scores32[:, j])
            if cand is not None:
                keep = min(self.keep, len(cand))
                best = cand[np.argpartition(-col[cand], keep - 1)[:keep]] if 0 < keep < len(cand) else cand
            else:
                keep = min(self.keep, rows)
                best = np.argpartition(-col, keep - 1)[:keep] if keep < rows else np.arange(rows)
            mine = self.returned[j]
            mine = mine[(mine >= base) & (mine < base + rows)] - base
            pos = np.unique(np.concatenate([best, mine]))
            t = scores_f64(chunk[pos], self.queries[j])
            self.truth[j].update(zip((pos + base).tolist(), t.tolist()))

    def for_query(self, j: int) -> Callable[[np.ndarray], np.ndarray]:
        table = self.truth[j]

        def referee(ordinals):
            return np.array([table[int(o)] for o in np.asarray(ordinals).tolist()], dtype=np.float64)
        referee.dim = int(self.queries.shape[1])
        return referee


def scores_full_chunked_refereed(chunks: Iterable[np.ndarray], queries: np.ndarray, returned: Sequence[Sequence[int]], keep: int,
                                 restrict: np.ndarray | None = None):
    """`scores_full_chunked` + a `ChunkedReferee` filled in the same pass over the chunks (`restrict`: the ordinals of a subset search)."""
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    ref = ChunkedReferee(queries, returned, keep, restrict)
    parts = []
    base = 0
    for chunk in chunks:
        sc = cosine_to_score(np.dot(chunk, queries.T))
        ref.see_chunk(base, chunk, sc)
        parts.append(sc.T)
        base += chunk.shape[0]
    return np.ascontiguousarray(np.concatenate(parts, axis=1), dtype=np.float32), ref
