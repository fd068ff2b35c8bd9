"""GPU suite (`-m gpu`, runs on the MI355X box): the HIP path, reached through the
drop-in VectorBase class and through the ctypes C ABI, against
  * the committed golden vectors (answers of the VERBATIM reference class),
  * the numpy oracle on seeded random inputs (differential, modulo fp32 near-ties; scores within 1e-5 --
    `oracle.vectorbase_oracle.check_topk_parity`: where a float64 referee is passed -- every check that has the rows at hand --
    a different row at a rank is accepted only within 2 * (measured noise of the reference + measured noise of the device)
    of the float64 truth of the reference's row, and the device's noise is capped by the reference's own; the remaining
    small isotropic cases use the constant 4 * 2^-24 on the reference's float32 scores),
  * size-independent properties at BASELINE.json's full size (1M x 1536).
Nothing here reads /root/reference."""

import asyncio
import hashlib

import numpy as np
import pytest

from oracle import vectorbase_oracle as vo
from tests.fakes import NullModel, create_test_embedding_model
from tests.synth import explicit_case_arrays, make_corpus, make_queries, subset_choice
from typeagent_py_amd import ScoredInt, TextEmbeddingIndexSettings, VectorBase, _native

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-5  # north_star: cosine scores within 1e-5 fp32


def new_vb(vectors=None, dtype="fp32") -> VectorBase:
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=dtype)
    if vectors is not None and len(vectors):
        vb.add_embeddings(None, np.ascontiguousarray(vectors, dtype=np.float32))
    return vb


def items_scores(res):
    assert all(isinstance(r, ScoredInt) for r in res)
    return [r.item for r in res], [r.score for r in res]


def assert_matches_expect(res, expect, ref_scores=None, k=None, min_score=0.0):
    """Exact item sequence + scores within tolerance; if the exact sequence differs, fall back to the
    near-tie-aware checker (needs ref_scores) so that only genuine fp32 ties are forgiven."""
    items, scores = items_scores(res)
    if items == expect["items"]:
        np.testing.assert_allclose(scores, expect["scores"], atol=SCORE_TOL, rtol=0)
        return True
    # The item sequence differs from the reference's: legal only inside groups of (near-)equal
    # float32 scores, whose order the reference leaves to numpy's introselect.  The checker
    # verifies exactly that (rank i must hold a row whose reference score is within 4*2^-24 of
    # the reference's i-th best score -- no referee here: the goldens carry scores, not rows --
    # nothing better omitted, scores within 1e-5).
    assert ref_scores is not None, f"items differ: {items[:8]}... vs {expect['items'][:8]}..."
    vo.check_topk_parity(ref_scores, items, scores, k, min_score)
    ref_sc = np.asarray(expect["scores"], dtype=np.float64)
    for a, b, sa in zip(items, expect["items"], expect["scores"]):
        if a != b:
            assert np.sum(np.abs(ref_sc - sa) <= vo.TIE_EPS) > 1 or len(items) == k, f"rank holding {b} replaced by {a} without a tie"
    return False


# --------------------------------------------------------------------------------------
# the library really is the thing that runs
# --------------------------------------------------------------------------------------
def test_native_library_loaded_and_device_is_gfx950():
    import torch

    assert torch.cuda.is_available()
    lib = _native.load_library()
    assert lib.tavb_version() == _native.ABI_VERSION
    assert _native.device_count() >= 1
    eng = _native.Engine(0)
    assert eng.get_option("compute_units") >= 64
    eng.close()
    with open("/proc/self/maps") as f:
        assert "libtavb.so" in f.read()


# --------------------------------------------------------------------------------------
# golden vectors
# --------------------------------------------------------------------------------------
def test_golden_explicit_cases(golden):
    for case in golden["explicit"]:
        v, q = explicit_case_arrays(case)
        kw = dict(case["args"])
        vb = new_vb(v)
        if v.shape[0] == 0:
            vb._set_embedding_size(len(q))
        name = case["name"]
        if "subset" in case:
            if case.get("raises") == "IndexError":
                with pytest.raises(IndexError):
                    vb.fuzzy_lookup_embedding_in_subset(q, case["subset"], **kw)
                continue
            res = vb.fuzzy_lookup_embedding_in_subset(q, case["subset"], **kw)
        elif "predicate_mod" in case:
            m, r = case["predicate_mod"]
            res = vb.fuzzy_lookup_embedding(q, predicate=lambda i: i % m == r, **kw)
        else:
            res = vb.fuzzy_lookup_embedding(q, **kw)
        items, scores = items_scores(res)
        exp = case["expect"]
        assert len(items) == len(exp["items"]), name
        np.testing.assert_allclose(scores, exp["scores"], atol=SCORE_TOL, rtol=0, err_msg=name)
        if items != exp["items"]:
            # only exactly-equal scores may be ordered differently (ours: ascending ordinal / position)
            for a, b, sa in zip(items, exp["items"], exp["scores"]):
                if a != b:
                    assert exp["scores"].count(sa) > 1, f"{name}: {items} vs {exp['items']}"
            assert sorted(items) == sorted(exp["items"]), name


def test_reference_known_answer_exact(golden):
    # reference tests/test_vectorbase.py:239-252: items [0,1,2], scores exactly [1.0, 0.5, 0.0]
    vb = new_vb(np.array([[1, 0], [0, 1], [-1, 0]], dtype=np.float32))
    res = vb.fuzzy_lookup_embedding(np.array([1.0, 0.0], dtype=np.float32), max_hits=3, min_score=0.0)
    assert [r.item for r in res] == [0, 1, 2]
    assert [r.score for r in res] == [1.0, 0.5, 0.0]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("idx", range(7))
def test_golden_seeded_cases(golden, idx):
    entry = golden["seeded"][idx]
    v, q = make_corpus(entry["n"], entry["d"], entry["seed"])
    assert _sha(v) == entry["corpus_sha256"]
    vb = new_vb(v)
    ref_scores = vo.scores_full(v, q)
    exact = 0
    for run in entry["runs"]:
        kw = dict(run["args"])
        k = 10 if kw["max_hits"] is None else kw["max_hits"]
        ms = 0.0 if kw["min_score"] is None else kw["min_score"]
        if run["kind"] == "full":
            res = vb.fuzzy_lookup_embedding(q, **kw)
            exact += assert_matches_expect(res, run["expect"], ref_scores, k, ms)
        elif run["kind"] == "subset":
            sub = subset_choice(entry["n"], run["subset_args"]["size"], run["subset_args"]["seed"])
            res = vb.fuzzy_lookup_embedding_in_subset(q, sub, **kw)
            exact += assert_matches_expect(res, run["expect"])
        elif run["kind"] == "full_f16_corpus_f32_query":
            vb16 = new_vb(v, dtype="fp16")
            res = vb16.fuzzy_lookup_embedding(q, **kw)
            v16 = v.astype(np.float16).astype(np.float32)
            exact += assert_matches_expect(res, run["expect"], vo.scores_full(v16, q), k, ms)
        elif run["kind"] == "full_f16_corpus_f16_query":
            vb16 = new_vb(v, dtype="fp16")
            q16 = q.astype(np.float16).astype(np.float32)
            res = vb16.fuzzy_lookup_embedding(q16, **kw)
            v16 = v.astype(np.float16).astype(np.float32)
            exact += assert_matches_expect(res, run["expect"], vo.scores_full(v16, q16), k, ms)
    assert exact >= 1 or entry["d"] == 1  # d=1: every score is exactly 0.0 or 1.0, all ties


@pytest.mark.slow
def test_golden_cfg2_one_million_rows(golden):
    """BASELINE config 2 at full size: 1M x 1536 fp32, single query, top-32 -- against the
    answer of the verbatim reference recorded in the goldens."""
    entry = next(e for e in golden["seeded"] if e["name"].startswith("cfg2_1m"))
    v, q = make_corpus(entry["n"], entry["d"], entry["seed"])
    assert _sha(q) == entry["query_sha256"]
    vb = new_vb()
    vb.deserialize(v)  # adopt by reference: no second 6 GB host copy
    for run in entry["runs"]:
        kw = dict(run["args"])
        res = vb.fuzzy_lookup_embedding(q, **kw)
        items, scores = items_scores(res)
        assert items == run["expect"]["items"]  # smallest top-33 gap 8.9e-7 >> fp32 noise
        np.testing.assert_allclose(scores, run["expect"]["scores"], atol=SCORE_TOL, rtol=0)
    # size-independent properties on the same corpus
    planted = vb.fuzzy_lookup_embedding(v[777_777], max_hits=1, min_score=0.0)
    assert planted[0].item == 777_777 and abs(planted[0].score - 1.0) <= 1e-6
    top64 = vb.fuzzy_lookup_embedding(q, max_hits=64, min_score=0.0)
    top32 = vb.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    assert [r.item for r in top64[:32]] == [r.item for r in top32]  # prefix property
    s = [r.score for r in top64]
    assert s == sorted(s, reverse=True)
    thr = top32[-1].score
    cut = vb.fuzzy_lookup_embedding(q, max_hits=64, min_score=thr)
    assert [r.item for r in cut] == [r.item for r in top64 if r.score >= np.float32(thr)]


# --------------------------------------------------------------------------------------
# differential vs the oracle
# --------------------------------------------------------------------------------------
def _diff_one(v, q, k, min_score, vb, ref_scores=None):
    if ref_scores is None:
        ref_scores = vo.scores_full(v, q)
    res = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=min_score)
    items, scores = items_scores(res)
    kk = 10 if k is None else k
    ms = 0.0 if min_score is None else min_score
    return vo.check_topk_parity(ref_scores, items, scores, kk, ms)


@pytest.mark.parametrize("d", [1, 2, 3, 5, 17, 64, 100, 384, 768, 1024, 1536, 1540, 3072])
def test_differential_dims(d):
    n = 3000 if d <= 1536 else 1200
    v, q = make_corpus(n, d, 1000 + d)
    vb = new_vb(v)
    sc = vo.scores_full(v, q)
    exact = total = 0
    for k, ms in [(None, None), (1, 0.0), (10, 0.0), (32, 0.0), (50, 0.5), (64, 0.0), (65, 0.0), (200, 0.0), (256, 0.3), (32, 0.53), (32, 0.99), (10, 1.5)]:
        rep = _diff_one(v, q, k, ms, vb, sc)
        exact += rep.exact_positions
        total += rep.k_returned
    if d >= 17:
        assert exact == total, "no near-ties expected in gaussian data at this size"


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 511, 1000, 4097, 20000])
def test_differential_row_counts(n):
    v, q = make_corpus(n, 1536, 2000 + n)
    vb = new_vb(v)
    sc = vo.scores_full(v, q)
    for k in (1, 10, 32, n, n + 1):
        if k <= 256:
            _diff_one(v, q, k, 0.0, vb, sc)


def test_large_k_paging_and_zero_quirk():
    v, q = make_corpus(1500, 96, 77)
    vb = new_vb(v)
    sc = vo.scores_full(v, q)
    for k in (257, 300, 700, 1500, 1501, 5000):
        res = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=0.45)
        items, scores = items_scores(res)
        vo.check_topk_parity(sc, items, scores, k, 0.45)
    everything = vb.fuzzy_lookup_embedding(q, max_hits=0, min_score=0.5)  # [-0:] quirk: all survivors, sorted
    ref = vo.lookup(v, q, 0, 0.5)
    assert len(everything) == len(ref) > 256
    vo.check_topk_parity(sc, *items_scores(everything), 0, 0.5)


def test_predicate_path():
    v, q = make_corpus(2500, 64, 78)
    vb = new_vb(v)
    for pred in (lambda i: i % 7 == 3, lambda i: i > 2400, lambda i: False, lambda i: True):
        ref = vo.lookup(v, q, 12, 0.4, pred)
        res = vb.fuzzy_lookup_embedding(q, max_hits=12, min_score=0.4, predicate=pred)
        items, scores = items_scores(res)
        assert items == [i for i, _ in ref]
        np.testing.assert_allclose(scores, [s for _, s in ref], atol=SCORE_TOL, rtol=0)
    assert vb.fuzzy_lookup_embedding(q, max_hits=0, min_score=0.0, predicate=lambda i: True) == []
    # like the reference (:193-199) the predicate sees EVERY survivor, once, in ascending ordinal order
    seen = []
    vb.fuzzy_lookup_embedding(q, max_hits=3, min_score=0.5, predicate=lambda i: seen.append(i) or (i % 2 == 0))
    assert seen == np.flatnonzero(vo.scores_full(v, q) >= np.float32(0.5)).tolist() and len(seen) > 500


@pytest.mark.parametrize("dtype,n,d", [("fp32", 300_000, 1536), ("fp16", 120_000, 1536), ("fp32", 5_000, 100), ("fp32", 700, 3)])
def test_all_survivors_in_one_pass(dtype, n, d):
    """max_hits beyond the fused selection, max_hits == 0 and the predicate path: ONE emit-all scan (not a scan per 256
    results), sorted on the host; whole-corpus and subset forms."""
    v, q = make_corpus(n, d, 7900 + d)
    vb = new_vb(v, dtype=dtype)
    vv = _f16(v) if dtype == "fp16" else v
    sc = vo.scores_full(vv, q)
    eng = vb.engine
    eng.profile_enable(True)
    eng.profile_reset()
    res = vb.fuzzy_lookup_embedding(q, max_hits=0, min_score=0.0)  # every row survives: n results
    assert eng.profile_read(_native.KERNEL_SCAN)[1] == 1
    assert len(res) == n
    vo.check_topk_parity(sc, *items_scores(res), 0, 0.0)
    res = vb.fuzzy_lookup_embedding(q, max_hits=3000, min_score=0.0)
    assert len(res) == min(3000, n)
    vo.check_topk_parity(sc, *items_scores(res), 3000, 0.0)
    thr = float(np.sort(sc)[-400]) if n > 1000 else 0.5
    res = vb.fuzzy_lookup_embedding(q, max_hits=0, min_score=thr)
    ref = vo.lookup(vv, q, 0, thr)
    assert len(res) == len(ref)
    vo.check_topk_parity(sc, *items_scores(res), 0, thr)
    sub = subset_choice(n, min(n, 2000), 7901) + [0, 0, -1]
    res = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=0, min_score=0.0)
    ref = vo.lookup_in_subset(vv, q, sub, 0, 0.0)
    assert len(res) == len(ref) == len(sub)
    got_sorted, ref_sorted = sorted((r.item, r.score) for r in res), sorted(ref)
    assert [i for i, _ in got_sorted] == [i for i, _ in ref_sorted]
    np.testing.assert_allclose([s_ for _, s_ in got_sorted], [s_ for _, s_ in ref_sorted], atol=SCORE_TOL, rtol=0)
    assert [r.score for r in res] == sorted((r.score for r in res), reverse=True)
    res = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=500, min_score=0.0)
    assert len(res) == min(500, len(sub))
    eng.profile_enable(False)


def test_subset_search_differential():
    v, q = make_corpus(5000, 1536, 79)
    vb = new_vb(v)
    rng = np.random.default_rng(5)
    for size, k, ms in [(1, 10, 0.0), (20, 32, 0.0), (1000, 10, 0.0), (3000, 50, 0.5), (700, 300, 0.0)]:
        sub = rng.choice(5000, size=size, replace=False).tolist()
        res = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=k, min_score=ms)
        items, scores = items_scores(res)
        sc = vo.cosine_to_score(np.dot(v[np.asarray(sub)], q))
        vo.check_topk_parity(sc, items, scores, k, ms, candidate_ordinals=np.asarray(sub))
    # duplicates and negative ordinals (numpy fancy-index semantics, vectorbase.py:217-229)
    sub = [4999, 17, 17, -1, 3, -5000]
    res = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=10, min_score=0.0)
    ref = vo.lookup_in_subset(v, q, sub, 10, 0.0)
    assert sorted(r.item for r in res) == sorted(i for i, _ in ref)
    np.testing.assert_allclose(sorted(r.score for r in res), sorted(s for _, s in ref), atol=SCORE_TOL, rtol=0)
    with pytest.raises(IndexError):
        vb.fuzzy_lookup_embedding_in_subset(q, [0, 5000])
    with pytest.raises(IndexError):
        vb.fuzzy_lookup_embedding_in_subset(q, [-5001])


def test_a_repeated_subset_keeps_its_row_list_on_the_device():
    """`fuzzy_lookup_embedding_in_subset` with the same list again (tools/benchmark_vectorbase.py:133-163: 1000 of 10k, seed 99, one list for
    every round; the memory provider's scope list, storage/memory/messageindex.py:173-183): the wrapped, range-checked row list is uploaded
    once (`tavb_search_subset_resident`); a list edited in place is a new subset.  Same answers as the plain path and the oracle."""
    v, q = make_corpus(10_000, 1536, 43)
    vb = new_vb(v)
    subset = np.random.default_rng(99).choice(10_000, size=1000, replace=False).tolist()
    first = vb.fuzzy_lookup_embedding_in_subset(q, subset, max_hits=10, min_score=0.0)
    cache = vb._subset_cache
    assert cache is not None and cache[0] is subset and tuple(cache[4].shape) == (1000,)
    ptr = cache[4].data_ptr()
    again = vb.fuzzy_lookup_embedding_in_subset(q, subset, max_hits=10, min_score=0.0)
    assert vb._subset_cache is cache and cache[4].data_ptr() == ptr
    assert [(r.item, r.score) for r in first] == [(r.item, r.score) for r in again]
    sub = np.asarray(subset, dtype=np.int64)
    vo.check_topk_parity(vo.scores_full(v, q)[sub], [r.item for r in first], [r.score for r in first], 10, 0.0, candidate_ordinals=sub,
                         referee=vo.f64_referee(v[sub], q))
    plain = vb.fuzzy_lookup_embedding_in_subset(q, tuple(subset), max_hits=10, min_score=0.0)  # (a tuple: the per-call upload path)
    assert [(r.item, r.score) for r in plain] == [(r.item, r.score) for r in first]
    best = first[0].item
    subset[subset.index(best)] = subset[0]  # the best row leaves the subset (in place): seen
    edited = vb.fuzzy_lookup_embedding_in_subset(q, subset, max_hits=10, min_score=0.0)
    assert best not in [r.item for r in edited] and vb._subset_cache is not cache
    sub = np.asarray(subset, dtype=np.int64)
    vo.check_topk_parity(vo.scores_full(v, q)[sub], [r.item for r in edited], [r.score for r in edited], 10, 0.0, candidate_ordinals=sub,
                         referee=vo.f64_referee(v[sub], q))
    # negative ordinals, duplicates, an ndarray; then the index grows and the same array means other rows
    arr = np.array([-1, 5, 5, 9_999, 17, -10_000], dtype=np.int64)
    a1 = vb.fuzzy_lookup_embedding_in_subset(v[9_999], arr, max_hits=3, min_score=0.0)
    assert [r.item for r in a1][:2] == [-1, 9_999] and abs(a1[0].score - 1.0) < 1e-6
    vb.add_embedding(None, q)
    a2 = vb.fuzzy_lookup_embedding_in_subset(q, arr, max_hits=3, min_score=0.0)
    assert a2[0].item == -1 and abs(a2[0].score - 1.0) < 1e-6  # -1 is the appended row now
    with pytest.raises(IndexError):
        vb.fuzzy_lookup_embedding_in_subset(q, [10_001], max_hits=3)


@pytest.mark.parametrize("nq", [1, 2, 3, 5, 8, 9, 20])
@pytest.mark.parametrize("d,k", [(1536, 32), (384, 50), (1536, 200), (10, 7)])
def test_batch_equals_sequential(nq, d, k):
    v, _ = make_corpus(4000, d, 300 + d)
    qs = make_queries(nq, d, 400 + nq)
    vb = new_vb(v)
    batch = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.3)
    assert len(batch) == nq
    for qi in range(nq):
        single = vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=0.3)
        bi, bs = items_scores(batch[qi])
        si, ss = items_scores(single)
        sc = vo.scores_full(v, qs[qi])
        vo.check_topk_parity(sc, bi, bs, k, 0.3)
        vo.check_topk_parity(sc, si, ss, k, 0.3)
        assert bi == si
        np.testing.assert_allclose(bs, ss, atol=2e-7, rtol=0)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_cursor_forms_enumerate_what_the_one_pass_form_returns(dtype):
    """`tavb_search_after` / `tavb_search_subset_after` (kept for callers that stop early): feeding the last hit of a page as the
    cursor of the next enumerates every survivor, best first, duplicates (exact ties) included -- the same sequence
    `tavb_search_all` / `tavb_search_subset_all` produce in one pass."""
    v, q = make_corpus(6_001, 384, 4300)
    v[100:140] = v[7]  # 41 rows with exactly equal scores: a page boundary falls inside the tie
    vb = new_vb(v, dtype=dtype)
    vb.fuzzy_lookup_embedding(q)  # sync the device mirror
    eng = vb.engine
    thr = np.float32(0.5)
    want_o, want_s = eng.search_all(v[7], thr)
    got_o, got_s, after = [], [], None
    while True:
        o, s = eng.search(v[7], 16, thr, after=after)
        if len(o) == 0:
            break
        got_o += o.tolist()
        got_s += s.tolist()
        after = (float(s[-1]), int(o[-1]))
        assert len(got_o) <= len(want_o)
    assert got_o == want_o.tolist() and got_s == want_s.tolist() and len(got_o) > 41
    rows = np.asarray(subset_choice(6_001, 900, 4301) + list(range(100, 140)) + [7, 7], dtype=np.int64)
    want_p, want_ps = eng.search_all(v[7], thr, subset_rows=rows)
    got_p, got_ps, after = [], [], None
    while True:
        p, s = eng.search_subset(v[7], rows, 10, thr, after=after)
        if len(p) == 0:
            break
        got_p += p.tolist()
        got_ps += s.tolist()
        after = (float(s[-1]), int(p[-1]))
        assert len(got_p) <= len(want_p)
    assert got_p == want_p.tolist() and got_ps == want_ps.tolist() and len(got_p) >= 43


def test_batch_lookup_as_arrays():
    v, _ = make_corpus(3_000, 384, 4100)
    qs = make_queries(70, 384, 4101)
    vb = new_vb(v)
    lists = vb.fuzzy_lookup_embeddings(qs, max_hits=7, min_score=0.55)
    o, s, c = vb.fuzzy_lookup_embeddings(qs, max_hits=7, min_score=0.55, as_arrays=True)
    assert o.shape == (70, 7) and s.dtype == np.float32 and c.dtype == np.int32
    for qi in range(70):
        assert [(r.item, r.score) for r in lists[qi]] == list(zip(o[qi, : c[qi]].tolist(), s[qi, : c[qi]].tolist()))
    e = new_vb().fuzzy_lookup_embeddings(qs, max_hits=7, as_arrays=True)
    assert e[0].shape == (70, 7) and not e[2].any()
    with pytest.raises(ValueError):
        vb.fuzzy_lookup_embeddings(qs, max_hits=1000, as_arrays=True)


@pytest.mark.parametrize("d", [8, 384, 1536, 2048])
def test_f16_corpus_differential(d):
    v, q = make_corpus(6000, d, 500 + d)
    vb = new_vb(v, dtype="fp16")
    v16 = v.astype(np.float16).astype(np.float32)  # what the device holds, widened
    sc = vo.scores_full(v16, q)
    for k, ms in [(10, 0.0), (32, 0.0), (200, 0.4)]:
        res = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=ms)
        vo.check_topk_parity(sc, *items_scores(res), k, ms)


@pytest.mark.parametrize("tier", [1, 2, 3])
def test_kernel_tiers_agree(tier):
    v, q = make_corpus(8192, 1536, 600)
    sc = vo.scores_full(v, q)
    vb = new_vb(v)
    vb.engine.set_option("force_tier", tier)
    res = vb.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    assert vb.engine.get_option("last_tier") == tier
    rep = vo.check_topk_parity(sc, *items_scores(res), 32, 0.0)
    assert rep.ordinals_bit_exact
    res = vb.fuzzy_lookup_embedding(q, max_hits=130, min_score=0.0)
    vo.check_topk_parity(sc, *items_scores(res), 130, 0.0)


@pytest.mark.parametrize("opts", [dict(scan_unroll=1), dict(scan_unroll=4), dict(scan_nt=0), dict(scan_pipe=1),
                                  dict(scan_pipe=1, scan_unroll=1), dict(scan_waves=4, scan_blocks=1024),
                                  dict(scan_waves=8, scan_blocks=512), dict(scan_waves=1, scan_blocks=7)])
def test_launch_geometry_variants_agree(opts):
    v, q = make_corpus(30000, 1536, 601)
    sc = vo.scores_full(v, q)
    vb = new_vb(v)
    for name, val in opts.items():
        vb.engine.set_option(name, val)
    res = vb.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    rep = vo.check_topk_parity(sc, *items_scores(res), 32, 0.0)
    assert rep.ordinals_bit_exact


def test_ties_resolve_to_ascending_ordinal():
    row = np.zeros(64, dtype=np.float32)
    row[3] = 1.0
    v = np.tile(row, (1000, 1))
    vb = new_vb(v)
    res = vb.fuzzy_lookup_embedding(row, max_hits=10, min_score=0.0)
    assert [r.item for r in res] == list(range(10)) and all(r.score == 1.0 for r in res)
    res = vb.fuzzy_lookup_embedding(row, max_hits=300, min_score=0.0)
    assert [r.item for r in res] == list(range(300))


def test_nan_zero_and_clip_rows():
    v = np.array([[0, 0], [np.nan, 1], [3, 0], [-3, 0], [0.5, 0], [np.inf, 0], [-np.inf, 0]], dtype=np.float32)
    vb = new_vb(v)
    with np.errstate(invalid="ignore"):
        ref = vo.lookup(v, np.array([1, 0], dtype=np.float32), 10, 0.0)
    res = vb.fuzzy_lookup_embedding(np.array([1.0, 0.0], dtype=np.float32), max_hits=10, min_score=0.0)
    got = sorted((r.item, r.score) for r in res)
    assert got == sorted(ref)  # NaN row dropped, +-inf clipped to 1/0, zero row scores exactly 0.5


def test_float32_threshold_rule():
    c = float(np.float32(0.85)) * 2 - 1
    below = np.nextafter(np.nextafter(np.float32(c), np.float32(-1)), np.float32(-1))  # score = 0.85f - 1ulp
    v = np.array([[c, 0], [below, 0]], dtype=np.float32)
    vb = new_vb(v)
    q = np.array([1.0, 0.0], dtype=np.float32)
    ref = vo.lookup(v, q, 10, 0.85)
    assert [i for i, _ in ref] == [0]
    res = vb.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.85)
    assert [r.item for r in res] == [i for i, _ in ref]
    ref64 = vo.lookup(v, q, 10, np.float64(0.85))
    res64 = vb.fuzzy_lookup_embedding(q, max_hits=10, min_score=np.float64(0.85))
    assert [r.item for r in res64] == [i for i, _ in ref64]


# --------------------------------------------------------------------------------------
# API behaviour that needs the device (reference tests/test_vectorbase.py:148-159, 209-236)
# --------------------------------------------------------------------------------------
def test_fuzzy_lookup_by_text_with_fake_model():
    vb = VectorBase(TextEmbeddingIndexSettings(create_test_embedding_model()))

    async def go():
        for key in ("word1", "word2", "word3"):
            await vb.add_key(key)
        return await vb.fuzzy_lookup("word1", max_hits=2, min_score=0.0)

    results = asyncio.run(go())
    assert 1 <= len(results) <= 2
    assert results[0].item == 0
    assert results[0].score > 0.9


def test_lookup_texts_batched_equals_sequential_fuzzy_lookup():
    """the batched form of TermEmbeddingIndex.lookup_terms (reference
    storage/memory/reltermsindex.py:320-332) == the sequential fuzzy_lookup loop."""
    from typeagent_py_amd.adapters import lookup_texts_batched

    vb = VectorBase(TextEmbeddingIndexSettings(create_test_embedding_model(embedding_size=48), min_score=0.3, max_matches=7))
    words = [f"term number {i} about {'cats' if i % 3 else 'dogs'}" for i in range(200)]

    async def go():
        await vb.add_keys(words)
        queries = ["term number 17 about cats", "dogs", "something else entirely", words[150]]
        batched = await lookup_texts_batched(vb, queries)
        sequential = [await vb.fuzzy_lookup(q) for q in queries]
        return batched, sequential, await vb.get_embeddings(queries)

    batched, sequential, embedded = asyncio.run(go())
    assert len(batched) == 4
    corpus = np.asarray(vb.serialize(), dtype=np.float32)
    for b, s, e in zip(batched, sequential, embedded):
        assert [r.item for r in b] == [r.item for r in s] and len(b) <= 7
        np.testing.assert_allclose([r.score for r in b], [r.score for r in s], atol=2e-7, rtol=0)
        # ... and against the oracle's restatement of what the reference's loop does per text (storage/memory/reltermsindex.py:320-337 ->
        # vectorbase.py:232-246 -> :163-190 with the settings' defaults: max_matches 7, min_score 0.3)
        want = vo.lookup(corpus, np.asarray(e, dtype=np.float32), 7, 0.3)
        vo.check_topk_parity(vo.scores_full(corpus, np.asarray(e, dtype=np.float32)), *items_scores(b), 7, 0.3, referee=vo.f64_referee(corpus, np.asarray(e, dtype=np.float32)))
        assert len(b) == len(want)
        np.testing.assert_allclose([r.score for r in b], [s_ for _, s_ in want], atol=SCORE_TOL, rtol=0)
    assert batched[0][0].item == 17 and batched[3][0].item == 150
    assert asyncio.run(lookup_texts_batched(vb, [])) == []


def test_message_rerank_on_device_matches_provider_semantics():
    """chunk rows -> message ordinals ON THE DEVICE (lookup + accept bitmap + per-message reduction in one
    submission), in the order of operations of each provider.  The expectation is oracle/messages_oracle.py, which is pinned
    to the verbatim `SqliteMessageTextIndex` in tests/test_reference_consumers.py."""
    from oracle import messages_oracle as mo
    from typeagent_py_amd.adapters import lookup_messages_by_embedding, lookup_messages_in_subset

    n = 30_000
    v, q = make_corpus(n, 384, 9400)
    rng = np.random.default_rng(9401)
    row_to_msg = np.sort(rng.integers(0, 9_000, size=n)).astype(np.int64)  # 1 .. ~10 chunks per message, some ordinals unused
    row_to_msg[rng.choice(n, size=200, replace=False)] = -1  # rows without a message (the SQL lookup finds nothing for them)
    vb = new_vb(v)
    look = lambda e, k, t: vo.lookup(v, e, k, t)
    qs = [q] + list(make_queries(4, 384, 9402))
    # near-duplicates of one query inside one message, so that "best score per message" has something to do
    big = int(row_to_msg[np.flatnonzero(row_to_msg >= 0)[500]])
    rows_big = np.flatnonzero(row_to_msg == big)
    for r in rows_big:
        w = qs[1] + 0.5 * rng.standard_normal(384).astype(np.float32) / np.sqrt(384)
        v[r] = w / np.linalg.norm(w)
    vb = new_vb(v)
    for qi, e in enumerate(qs):
        for k, t, subset in ((25, 0.5, None), (None, None, None), (200, 0.0, None), (25, 0.0, list(range(0, 9_000, 2))), (40, 0.45, [big, 7, 7, 8999, 12345678]),
                             (10, 0.0, [])):
            want = mo.sqlite_lookup_by_embedding(look, e, row_to_msg, k, t, subset)
            got = lookup_messages_by_embedding(vb, e, row_to_msg, max_matches=k, threshold_score=t, accept=subset)
            assert [h.item for h in got] == [m for m, _ in want], (qi, k, t)
            np.testing.assert_allclose([h.score for h in got], [s for _, s in want], atol=SCORE_TOL, rtol=0)
    got = lookup_messages_by_embedding(vb, qs[1], row_to_msg, max_matches=25, threshold_score=0.0)
    assert got[0].item == big and sum(1 for h in got if h.item == big) == 1 and len(got) < 25  # several chunks of `big` collapsed
    # an arbitrary callable predicate: lookup on the device, aggregation on the host, same answer as the collection form
    even = lookup_messages_by_embedding(vb, qs[0], row_to_msg, 25, 0.0, accept=lambda m: m % 2 == 0)
    assert [h.item for h in even] == [h.item for h in lookup_messages_by_embedding(vb, qs[0], row_to_msg, 25, 0.0, accept=list(range(0, 9_000, 2)))]
    # memory provider: true subset gather over index positions, then best score per message (no -1 rows there)
    rtm = np.where(row_to_msg < 0, 0, row_to_msg)
    subset = list(range(100, 4000)) + rows_big.tolist()
    for e in qs[:3]:
        hits = vo.lookup_in_subset(v, e, subset, 10, 0.0)
        want = mo.memory_messages_from_hits(hits, rtm)
        got = lookup_messages_in_subset(vb, e, subset, rtm, max_matches=10, threshold_score=0.0)
        assert [h.item for h in got] == [m for m, _ in want]
    # the map must cover the index
    vb.add_embeddings(None, v[:3])
    with pytest.raises(ValueError, match="covers"):
        vb.lookup_messages_by_embedding(qs[0], 5, 0.0)


def test_subset_semantics_from_reference_tests():
    vb = VectorBase(TextEmbeddingIndexSettings(create_test_embedding_model()))
    samples = [np.array(x, dtype=np.float32) for x in ([0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.7, 0.8, 0.9])]
    for s in samples:
        vb.add_embedding(None, s)
    result = vb.fuzzy_lookup_embedding_in_subset(samples[0], [0, 1, 2])
    assert len(result) > 0 and 0 in [r.item for r in result]
    result = vb.fuzzy_lookup_embedding_in_subset(samples[0], [1])
    assert len(result) == 1 and result[0].item == 1
    assert vb.fuzzy_lookup_embedding_in_subset(samples[0], []) == []


def test_appends_between_lookups_keep_device_in_sync():
    v, q = make_corpus(5000, 384, 900)
    vb = new_vb(v[:10])
    for upto in (10, 11, 100, 1000, 5000):
        if len(vb) < upto:
            if upto - len(vb) == 1:
                vb.add_embedding(None, v[len(vb)])
            else:
                vb.add_embeddings(None, v[len(vb):upto])
        res = vb.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.0)
        vo.check_topk_parity(vo.scores_full(v[:upto], q), *items_scores(res), 10, 0.0, referee=vo.f64_referee(v[:upto], q))
    vb.clear()
    assert vb.fuzzy_lookup_embedding(q) == []
    vb.add_embeddings(None, v[100:200])
    res = vb.fuzzy_lookup_embedding(q, max_hits=5, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(v[100:200], q), *items_scores(res), 5, 0.0, referee=vo.f64_referee(v[100:200], q))
    other = np.ascontiguousarray(v[300:400])
    vb.deserialize(other)
    res = vb.fuzzy_lookup_embedding(q, max_hits=5, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(other, q), *items_scores(res), 5, 0.0, referee=vo.f64_referee(other, q))


def test_in_place_edit_of_the_serialized_matrix_is_noticed():
    """serialize() hands out the live host matrix (like the reference, :271).  Editing it in place used to leave the device
    mirror stale without a word; the sampled fingerprint now notices (bulk edits) and mark_dirty() stays the explicit form."""
    v, q = make_corpus(5000, 256, 4400)
    vb = new_vb(v)
    first = vb.fuzzy_lookup_embedding(q, max_hits=5, min_score=0.0)
    live = vb.serialize()
    live[:] = np.roll(live, 7, axis=0)  # every row moves 7 places down
    moved = vb.fuzzy_lookup_embedding(q, max_hits=5, min_score=0.0)
    assert [r.item for r in moved] == [(r.item + 7) % 5000 for r in first]
    x = np.ascontiguousarray(v[::-1])
    vb.deserialize(x)  # kept by reference (:287): the caller still holds x
    x *= -1.0
    flipped = vb.fuzzy_lookup_embedding(q, max_hits=3, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(x, q), *items_scores(flipped), 3, 0.0, referee=vo.f64_referee(x, q))


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_small_corpus_lookups_replay_a_captured_graph(dtype):
    """At the reference's own scale (10k x 1536: 376-417 us per lookup there) a lookup here is launch-bound; single-query lookups on small
    corpora can replay ONE captured HIP graph (query H2D + scan + merge into pinned host memory; option `graph_max_bytes`).  Same
    answers as the plain path, whatever is replayed: other queries, other (k, min_score) shapes, appends, the profiler on."""
    v, _ = make_corpus(10_000, 1536, 4500)
    qs = make_queries(12, 1536, 4501)
    vb = new_vb(v, dtype=dtype)
    eng = vb.engine
    assert eng.get_option("graph_max_bytes") == 0  # opt-in: the replay measured slower than three plain submissions on ROCm 7.2
    eng.set_option("graph_max_bytes", 256 << 20)
    seen = _f16(v) if dtype == "fp16" else v
    replayed = 0
    for rep in range(3):
        for qi in range(12):
            k, ms = (10, 0.0) if qi % 3 else (32, 0.5)
            res = vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=ms)
            replayed += eng.get_option("last_graph")
            vo.check_topk_parity(vo.scores_full(seen, qs[qi]), *items_scores(res), k, ms, referee=vo.f64_referee(seen, qs[qi]))
    assert replayed >= 30  # everything after the first two calls of each of the two shapes
    eng.set_option("graph_max_bytes", 0)
    plain = vb.fuzzy_lookup_embedding(qs[1], max_hits=10, min_score=0.0)
    assert eng.get_option("last_graph") == 0
    eng.set_option("graph_max_bytes", 256 << 20)
    again = vb.fuzzy_lookup_embedding(qs[1], max_hits=10, min_score=0.0)
    assert eng.get_option("last_graph") == 1 and [(r.item, r.score) for r in again] == [(r.item, r.score) for r in plain]
    # five shapes through four slots, then back to the first
    for k in (1, 2, 3, 4, 5, 1, 1, 1):
        for _ in range(3):
            res = vb.fuzzy_lookup_embedding(qs[2], max_hits=k, min_score=0.0)
        vo.check_topk_parity(vo.scores_full(seen, qs[2]), *items_scores(res), k, 0.0, referee=vo.f64_referee(seen, qs[2]))
    assert eng.get_option("last_graph") == 1
    # an append changes the corpus: the new rows are seen (a new shape: plain, capture, replay)
    vb.add_embedding(None, qs[3])
    for _ in range(3):
        res = vb.fuzzy_lookup_embedding(qs[3], max_hits=10, min_score=0.0)
        assert res[0].item == 10_000 and abs(res[0].score - 1.0) < 2e-3
    # with the profiler on the launches are timed one by one: no graph
    eng.profile_enable(True)
    vb.fuzzy_lookup_embedding(qs[3], max_hits=10, min_score=0.0)
    assert eng.get_option("last_graph") == 0
    eng.profile_enable(False)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("n", [1, 100, 1294, 10_000])
def test_small_corpus_lookups_take_one_launch(n, dtype):
    """The scale typeagent itself runs at (the reference's only real fixture is 1294 x 1536; its benchmark 10k x 1536): a single-query
    lookup on a corpus of up to 128 MiB is ONE launch -- the scan's per-workgroup lists land in pinned host memory and are merged on the
    host (tavb_merge_keys_host), no merge kernel.  Same answers as the two-launch path for every (k, min_score) the consumers use
    (k = 10 default, 50 @ 0.85 related terms, 25 @ 0.7 messages), incl. k > rows, k = 256, exact ties, and through the drop-in class."""
    v, q = make_corpus(n, 1536, 8300 + n)
    if n >= 100:
        v[7] = v[3]  # an exact tie
        q = (v[3] + 0.05 * make_queries(1, 1536, 9)[0]).astype(np.float32)
        q /= np.linalg.norm(q)
    vb = new_vb(v, dtype=dtype)
    eng = vb.engine
    seen = _f16(v) if dtype == "fp16" else v
    eng.profile_enable(True)
    for k, ms in [(10, 0.0), (50, 0.85), (25, 0.7), (1, 0.0), (256, 0.0), (32, 0.5)]:
        eng.profile_reset()
        res = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=ms)
        vo.check_topk_parity(vo.scores_full(seen, q), *items_scores(res), k, ms, referee=vo.f64_referee(seen, q))
        if k == 256 and n == 10_000:
            # the list budget (8192 keys) would cut the grid to 32 of the 256 workgroups for 31 / 61 MB of rows: the one-launch path steps aside
            # (less than half of the full grid left) and the full grid + the merge kernel serve the lookup
            assert eng.get_option("last_direct") == 0
            assert eng.profile_read(_native.KERNEL_SCAN)[1] == 1 and eng.profile_read(_native.KERNEL_MERGE)[1] == 1
            continue
        assert eng.get_option("last_direct") == 2  # one launch, and the 1536-wide query rode in its kernel arguments: no copy in front of it
        assert eng.profile_read(_native.KERNEL_SCAN)[1] == 1 and eng.profile_read(_native.KERNEL_MERGE)[1] == 0  # one launch
        eng.set_option("inline_query", 0)  # the same launch with the query copied into a device buffer first
        copied = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=ms)
        assert eng.get_option("last_direct") == 1 and items_scores(copied) == items_scores(res)
        eng.set_option("inline_query", 1)
        eng.set_option("small_direct_bytes", 0)
        two = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=ms)
        assert eng.get_option("last_direct") == 0
        eng.set_option("small_direct_bytes", 128 << 20)
        assert items_scores(res) == items_scores(two)  # the same kernel, the same keys: the host merge and the merge kernel agree exactly
    if n >= 100:
        res = vb.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.0)
        assert [r.item for r in res[:2]] == [3, 7]  # ties by ascending ordinal
    # a launch geometry forced by option is respected up to the list budget
    eng.set_option("scan_blocks", 16)
    res = vb.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(seen, q), *items_scores(res), 10, 0.0, referee=vo.f64_referee(seen, q))
    eng.set_option("scan_blocks", 0)
    eng.profile_enable(False)
    # a scan form that has no inline-query variant copies the query as before; consecutive lookups with different queries see their own query
    eng.set_option("scan_unroll", 4)
    res = vb.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.0)
    assert eng.get_option("last_direct") == 1
    vo.check_topk_parity(vo.scores_full(seen, q), *items_scores(res), 10, 0.0, referee=vo.f64_referee(seen, q))
    eng.set_option("scan_unroll", 2)
    for seed in range(5):
        q2 = make_queries(1, 1536, 50 + seed)[0]
        res = vb.fuzzy_lookup_embedding(q2, max_hits=10, min_score=-1.0)
        assert eng.get_option("last_direct") == 2
        vo.check_topk_parity(vo.scores_full(seen, q2), *items_scores(res), 10, -1.0, referee=vo.f64_referee(seen, q2))
    # SEVERAL queries at once (batched related-term lookups) take it too, in its grouped form (end of round 6): ONE launch of (row workgroups) x
    # (query groups), lists merged on the host per query -- and the answers are the single lookups' bit for bit (the same summation order); with
    # the grouped form off, 2 .. 8 queries take the plain multi-query launch or the tiles as before and 9+ the tiles; subset lookups are unaffected
    qs = np.concatenate([q[None, :], make_queries(63, 1536, 77)])
    if n >= 100:
        qs[3] = v[min(n - 1, 50)]
    singles = {}

    def single(qi, k, ms):
        if (qi, k, ms) not in singles:
            singles[(qi, k, ms)] = items_scores(vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=ms))
        return singles[(qi, k, ms)]

    for nq in (2, 3, 4, 8, 9, 33, 64):
        for k, ms in [(10, 0.0), (50, 0.85), (50, -1.0), (200, 0.0)]:
            if nq > 9 and k == 200 and n == 10_000:
                continue  # (the sequential lookups of the comparison below are the slow part)
            eng.profile_enable(True)
            eng.profile_reset()
            out = vb.fuzzy_lookup_embeddings(qs[:nq], max_hits=k, min_score=ms)
            direct = eng.get_option("last_direct")
            launches = eng.profile_read(_native.KERNEL_SCAN)[1], eng.profile_read(_native.KERNEL_MERGE)[1]
            eng.profile_enable(False)
            if n <= 1294 and k <= 50:
                assert direct == 3, (nq, k, direct)  # (bigger shapes: wherever the cost model expects it to beat the tiles, plan_direct_group)
            if direct:
                assert launches == (1, 0)
            for qi in range(nq):
                if direct == 3:
                    assert items_scores(out[qi]) == single(qi, k, ms), (nq, k, ms, qi)
                if qi < 3 or qi == nq - 1:
                    vo.check_topk_parity(vo.scores_full(seen, qs[qi]), *items_scores(out[qi]), k, ms, referee=vo.f64_referee(seen, qs[qi]))
            if k == 200:
                continue
            max_nq = eng.get_option("direct_group_max_nq")
            eng.set_option("direct_group_max_nq", 0)
            old = vb.fuzzy_lookup_embeddings(qs[:nq], max_hits=k, min_score=ms)
            direct = eng.get_option("last_direct")
            eng.set_option("direct_group_max_nq", max_nq)
            if nq <= 4 and k <= 50 and n <= 1294:
                assert direct == 1, (nq, k, direct)  # (bigger shapes: when the list budget still covers the rows in two rounds of the grid)
            if nq >= 9:
                assert direct == 0
            for qi in (0, nq - 1):  # (the tiles: another summation order, the last bit of a score may differ)
                np.testing.assert_allclose([r.score for r in old[qi]], [r.score for r in out[qi]], atol=1e-6, rtol=0)
    # every group size gives the same answers (the option forces the form whatever the cost model says)
    ref = vb.fuzzy_lookup_embeddings(qs[:13], max_hits=10, min_score=0.0)
    for group in (1, 2, 4, 8):
        eng.set_option("direct_group", group)
        got = vb.fuzzy_lookup_embeddings(qs[:13], max_hits=10, min_score=0.0)
        assert eng.get_option("last_direct") == 3
        assert [items_scores(r) for r in got] == [items_scores(r) for r in ref], group
    eng.set_option("direct_group", 0)
    # per-query thresholds through the C ABI's batch call
    thr = np.array([0.0, 0.9, -1.0, 0.5], dtype=np.float32)
    o, s_, c_ = eng.search_batch(qs[:4], 10, thr)
    assert eng.get_option("last_direct") == 3
    for qi in range(4):
        vo.check_topk_parity(vo.scores_full(seen, qs[qi]), o[qi, : c_[qi]].tolist(), s_[qi, : c_[qi]].tolist(), 10, float(thr[qi]), referee=vo.f64_referee(seen, qs[qi]))
    thr = np.linspace(0.0, 0.9, 40).astype(np.float32)  # ... and a grouped launch of many groups reads each query's own threshold
    o, s_, c_ = eng.search_batch(qs[:40], 10, thr)
    assert eng.get_option("last_direct") == (3 if n <= 1294 else eng.get_option("last_direct"))
    for qi in (0, 7, 22, 39):
        vo.check_topk_parity(vo.scores_full(seen, qs[qi]), o[qi, : c_[qi]].tolist(), s_[qi, : c_[qi]].tolist(), 10, float(thr[qi]), referee=vo.f64_referee(seen, qs[qi]))
    # the device-resident form (tavb_search_device: sharded and fused paths): the grouped scan + ONE merge launch, the same keys
    import torch

    dq = torch.from_numpy(qs[:24]).cuda()
    eng.profile_enable(True)
    eng.profile_reset()
    keys = eng.search_device(dq, 10, 0.0)
    eng.synchronize()
    if n <= 1294:
        assert eng.get_option("last_direct") == 4
        assert (eng.profile_read(_native.KERNEL_SCAN)[1], eng.profile_read(_native.KERNEL_MERGE)[1]) == (1, 1)
    eng.profile_enable(False)
    ords, scs, cnts = _native.decode_keys(keys.cpu().numpy())
    host = eng.search_batch(qs[:24], 10, np.float32(0.0))
    if eng.get_option("last_direct") == 3 and n <= 1294:
        for qi in range(24):
            m = int(cnts[qi])
            assert m == int(host[2][qi]) and ords[qi, :m].tolist() == host[0][qi, :m].tolist() and scs[qi, :m].tolist() == host[1][qi, :m].tolist()


def test_wrong_query_size_raises_value_error():
    vb = new_vb(np.ones((4, 8), dtype=np.float32))
    with pytest.raises(ValueError):
        vb.fuzzy_lookup_embedding(np.ones(7, dtype=np.float32))


# --------------------------------------------------------------------------------------
# MFMA batched path on fp16 corpora: the 256-query fp16 tile is an exact FILTER (64 candidates per query by fp16-query
# score), the candidates are rescored with the fp32 query (tavb_rescore.hip): same meaning as the single-query kernels
# --------------------------------------------------------------------------------------
def _f16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("n,nq,k,ms,splits", [
    (20_000, 40, 32, 0.0, 0),
    (20_000, 300, 10, 0.52, 0),
    (5_000, 64, 48, 0.0, 3),
    (100, 33, 32, 0.0, 0),
    (70_001, 256, 32, 0.0, 17),
    (33_000, 1024, 32, 0.0, 0),
    (9_000, 50, 5, 0.9, 8),
    (30_000, 700, 32, 0.0, 0),  # three query tiles: 80 row ranges, 240 workgroups
])
def test_mfma_batch_against_oracle(n, nq, k, ms, splits):
    v, _ = make_corpus(n, 1536, 7000 + n % 97)
    qs = make_queries(nq, 1536, 7100 + nq)
    qs[0] = v[n // 2]  # plant an exact match
    vb = new_vb(v, dtype="fp16")
    eng = vb.engine
    eng.set_option("mfma_min_batch", 32)
    eng.set_option("mfma_splits", splits)
    eng.set_option("direct_group_max_nq", 0)  # (a batch of up to 128 queries on a corpus this small is the grouped streaming launch's otherwise)
    eng.profile_enable(True)
    eng.profile_reset()
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms)
    ms_mfma, n_mfma = eng.profile_read(_native.KERNEL_MFMA)
    assert n_mfma == 1, "the MFMA kernel must be the one that ran"
    assert eng.profile_read(_native.KERNEL_RESCORE)[1] >= 2 and eng.get_option("last_flagged") == 0
    v16 = _f16(v)  # the corpus values the device multiplies; the queries stay fp32
    exact = total = 0
    check = range(nq) if nq <= 64 else list(range(0, nq, max(1, nq // 48))) + [nq - 1]
    for qi in check:
        sc = vo.scores_full(v16, qs[qi])
        items, scores = items_scores(out[qi])
        rep = vo.check_topk_parity(sc, items, scores, k, ms)
        exact += rep.exact_positions
        total += rep.k_returned
    assert out[0][0].item == n // 2 and abs(out[0][0].score - 1.0) < 2e-3
    assert exact >= total - 2  # gaussian data: (near-)ties are vanishingly rare


def _ladder_phases(rows: int, sample: int, growth: int) -> int:
    """Phase count of the threshold ladder (tavb_abi.hip, tavb_search_device_dispatch)."""
    sample = (sample + 255) // 256 * 256
    if sample <= 0 or rows < 8 * sample:
        return 1
    bounds, done = 2, sample
    while growth > 0 and done * (growth + 1) * 2 <= rows and bounds < 8:
        done += done * growth
        bounds += 1
    return bounds


@pytest.mark.parametrize("n,nq,k,ms,sample,ladder", [(70_001, 300, 10, 0.52, 4096, 0), (70_001, 300, 10, 0.52, 2048, 4), (40_000, 64, 32, 0.0, 2048, 0),
                                                      (33_000, 1024, 32, 0.0, 1024, 1), (20_000, 40, 48, 0.0, 256, 4), (150_000, 256, 32, 0.0, 512, 2)])
def test_mfma_threshold_ladder_does_not_change_results(n, nq, k, ms, sample, ladder):
    """The phases of the threshold ladder (every row scanned once; the k-th best so far, a valid lower bound on every
    query's final k-th best score, seeds the next phase's admission test) must not change any answer: same
    ordinals/scores as a single un-seeded pass, and as the oracle."""
    v, _ = make_corpus(n, 1536, 9300 + n % 91)
    qs = make_queries(nq, 1536, 9301 + nq)
    qs[0] = v[5]       # best hit inside the first phase
    qs[1] = v[n - 3]   # best hit in the last phase
    qs[2] = v[sample + 7]  # best hit right behind the first boundary
    vb = new_vb(v, dtype="fp16")
    eng = vb.engine
    eng.set_option("mfma_min_batch", 32)
    eng.set_option("mfma_sample_rows", sample)
    eng.set_option("mfma_ladder", ladder)
    eng.profile_enable(True)
    eng.profile_reset()
    with_pass = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms)
    phases = _ladder_phases(n, sample, ladder)
    assert phases >= 2 and (ladder == 0) == (phases == 2)
    assert eng.profile_read(_native.KERNEL_MFMA_SAMPLE)[1] == phases - 1 and eng.profile_read(_native.KERNEL_MFMA)[1] == 1
    eng.set_option("mfma_sample_rows", -1)
    without = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms)
    v16 = _f16(v)
    for qi in range(nq):
        assert [(r.item, r.score) for r in with_pass[qi]] == [(r.item, r.score) for r in without[qi]]
    for qi in list(range(0, nq, max(1, nq // 24))) + [1, 2]:
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(with_pass[qi]), k, ms, referee=vo.f64_referee(v16, qs[qi]))
    assert with_pass[0][0].item == 5 and with_pass[1][0].item == n - 3 and with_pass[2][0].item == sample + 7


@pytest.mark.parametrize("n,nq", [(30_720, 1024), (50_003, 1024), (81_919, 300), (150_001, 128), (31_000, 65)])
def test_small_corpus_takes_two_default_phases_with_the_same_answers(n, nq):
    """Round 6: with the DEFAULT options a corpus of 30720 rows and more, but below eight first phases' worth, is scanned in two phases (an
    eighth of the rows in whole tiles, then the rest) instead of one un-seeded pass -- tavb_plan_ladder says so, the profile counters agree,
    and every answer is that of the single pass and of the oracle."""
    k = 32
    v, _ = make_corpus(n, 1536, 9400 + n % 89)
    qs = make_queries(nq, 1536, 9401 + nq)
    bounds = _native.plan_ladder(n, nq)
    assert len(bounds) == 3 and bounds[1] == n // 8 // 320 * 320
    qs[0] = v[3]               # best hit inside the first phase
    qs[1] = v[n - 2]           # ... in the last rows
    qs[2] = v[bounds[1]]       # ... in the first row behind the boundary
    vb = new_vb(v, dtype="fp16")
    eng = vb.engine
    eng.profile_enable(True)
    eng.profile_reset()
    two = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert eng.get_option("last_tier") == 4
    assert eng.profile_read(_native.KERNEL_MFMA_SAMPLE)[1] == 1 and eng.profile_read(_native.KERNEL_MFMA)[1] == 1
    eng.set_option("mfma_sample_rows", -1)
    one = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    v16 = _f16(v)
    for qi in range(nq):
        assert [(r.item, r.score) for r in two[qi]] == [(r.item, r.score) for r in one[qi]]
    for qi in list(range(0, nq, max(1, nq // 16))) + [1, 2]:
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(two[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
    assert two[0][0].item == 3 and two[1][0].item == n - 2 and two[2][0].item == bounds[1]


def test_mfma_tile_against_the_exact_tile_on_a_large_corpus():
    """Two independent matrix-core kernels over 786k rows x 1024 queries: the 256-query tile (4 waves, inline-asm MFMAs with
    AGPR+VGPR accumulators, buffer-descriptor LDS-DMA of whole lines, select kernel) + fp32 rescoring, against the 64-query
    tile fed split hi/lo query planes (builtin MFMAs, flat LDS-DMA, per-workgroup lists + merge kernel): every returned
    ordinal identical, scores within fp32 summation noise.  The asm MFMAs are invisible to the compiler's hazard recognizer,
    so this is the test that would notice a scheduling change breaking them; several ladder phases, partial last tile."""
    import torch

    n, nq, k = 786_432 + 77, 1024, 32
    eng = _native.Engine(0)
    corpus = torch.empty((n, 1536), dtype=torch.float16, device="cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4242)
    for lo in range(0, n, 131072):
        hi = min(n, lo + 131072)
        tmp = torch.empty((hi - lo, 1536), dtype=torch.float32, device="cuda")
        tmp.normal_(generator=gen)
        eng.normalize_rows_(tmp)
        corpus[lo:hi].copy_(eng.to_f16(tmp))
    eng.set_corpus_tensor(corpus)
    dq = torch.from_numpy(make_queries(nq, 1536, 4243)).cuda()
    eng.set_option("mfma_sample_rows", 16384)  # several ladder phases
    out = eng.search_device(dq, k, 0.0)
    eng.synchronize()
    assert eng.get_option("last_tier") == 4 and eng.get_option("last_flagged") == 0
    o_w, s_w, c_w = _native.decode_keys(out.cpu().numpy())
    eng.set_option("mfma_min_batch", 1 << 30)  # the same batch on the 64-query tile (16 query tiles per row range)
    eng.set_option("mfma_min_batch_big", 1 << 30)  # (this corpus is 2.4 GB: the big-corpus threshold too)
    out = eng.search_device(dq, k, 0.0)
    eng.synchronize()
    assert eng.get_option("last_tier") == 5
    o_s, s_s, c_s = _native.decode_keys(out.cpu().numpy())
    np.testing.assert_array_equal(c_w, c_s)
    np.testing.assert_allclose(s_w, s_s, atol=3e-7, rtol=0)
    assert np.all(c_w == k) and np.all(np.diff(s_w, axis=1) <= 0)
    # two different fp32 summation orders: the ordinal sequences may differ only by swaps inside fp32 near-ties
    diff = np.argwhere(o_w != o_s)
    assert len(diff) <= 16, len(diff)
    for qi, c in diff.tolist():  # (scores already agree to 3e-7 position by position)
        assert c == k - 1 or o_w[qi, c] in o_s[qi, max(0, c - 2) : c + 3], (qi, c)
    eng.close()


@pytest.mark.parametrize("nq", [64, 65, 100, 128, 129, 256, 1024])
def test_batch_equals_sequential_for_arbitrary_fp32_queries_on_fp16_corpus(nq):
    """`fuzzy_lookup_embeddings(E) == [fuzzy_lookup_embedding(e) for e in E]` at every batch size, for queries that are NOT
    fp16-representable: 64 rides the 64-query tile (split hi/lo planes), 65 .. 128 the 128-query tile and 129+ the 256-query
    tile, both + fp32 rescoring.  The
    sequential answers come from the streaming kernel (fp32 query x fp16 row), both are checked against the oracle."""
    v, _ = make_corpus(30_011, 1536, 7200)
    qs = make_queries(nq, 1536, 7201)
    assert np.any(_f16(qs) != qs)
    vb = new_vb(v, dtype="fp16")
    batch = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert vb.engine.get_option("last_tier") == (4 if nq >= 65 else 5)
    v16 = _f16(v)
    sample = sorted(set(np.linspace(0, nq - 1, 24).astype(int).tolist()))
    for qi in sample:
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=32, min_score=0.0)
        assert vb.engine.get_option("last_tier") in (1, 2, 3)
        assert [r.item for r in batch[qi]] == [r.item for r in seq]
        np.testing.assert_allclose([r.score for r in batch[qi]], [r.score for r in seq], atol=3e-7, rtol=0)
        rep = vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(batch[qi]), 32, 0.0, referee=vo.f64_referee(v16, qs[qi]))
        assert rep.ordinals_bit_exact
    # a threshold close to the scores: the relaxed filter threshold + exact re-test give the sequential counts
    thr = float(np.float32(batch[0][20].score))
    batch_t = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=thr)
    for qi in sample[:8]:
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=32, min_score=thr)
        assert [r.item for r in batch_t[qi]] == [r.item for r in seq]
    assert len(batch_t[0]) == 21

@pytest.mark.parametrize("n,nq,k,ms,sample", [(200_003, 100, 32, 0.0, 0), (120_000, 300, 10, 0.5, 4096), (641, 128, 48, 0.0, -1), (90_000, 65, 32, 0.0, 2048)])
def test_128_and_256_query_tiles_agree(n, nq, k, ms, sample):
    """The wide fp16 kernel comes in two widths (4 or 2 MFMA blocks per wave along the queries).  Same products, same
    candidates, same rescoring: the answers must be identical key for key, whatever the width, the number of query tiles
    (300 queries = 3 tiles of 128 or 2 of 256) and the ladder phases -- and equal to the oracle's."""
    v, _ = make_corpus(n, 1536, 8800 + nq)
    qs = make_queries(nq, 1536, 8801 + nq)
    qs[1] = v[n - 1]
    vb = new_vb(v, dtype="fp16")
    eng = vb.engine
    eng.set_option("mfma_sample_rows", sample)
    eng.profile_enable(True)
    res = {}
    for tile in (128, 256):
        eng.set_option("mfma_tile", tile)
        eng.profile_reset()
        res[tile] = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms)
        assert eng.get_option("last_tier") == 4 and eng.profile_read(_native.KERNEL_MFMA)[1] == 1
    for qi in range(nq):
        assert [(r.item, r.score) for r in res[128][qi]] == [(r.item, r.score) for r in res[256][qi]]
    eng.set_option("mfma_tile", 0)
    auto = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms)
    assert [(r.item, r.score) for r in auto[1]] == [(r.item, r.score) for r in res[128][1]] and auto[1][0].item == n - 1
    v16 = _f16(v)
    for qi in sorted(set(np.linspace(0, nq - 1, 16).astype(int).tolist())):
        rep = vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(res[128][qi]), k, ms, referee=vo.f64_referee(v16, qs[qi]))
        assert rep.tie_permuted_positions <= 2
    with pytest.raises(ValueError):
        eng.set_option("mfma_tile", 64)


def _plant_near_duplicates(v, qs, qi, rows, rng, eps=2e-4):
    """rows of v <- tiny perturbations of one vector near query qi: their scores to that query lie within ~1e-5 of one another"""
    base = qs[qi] + 0.3 * rng.standard_normal(v.shape[1]).astype(np.float32) / np.sqrt(v.shape[1])
    base /= np.linalg.norm(base)
    for r in rows:
        w = base + eps * rng.standard_normal(v.shape[1]).astype(np.float32) / np.sqrt(v.shape[1])
        v[r] = w / np.linalg.norm(w)


def test_wide_tile_band_holds_a_cluster_of_near_duplicates():
    """300 near-duplicate rows around rank k: the fp16-query scores cannot order them, so the wide tile keeps the whole BAND (every row
    within 2 delta of the approximate k-th best) and the fp32 rescoring of the band gives the exact answer -- no fallback pass."""
    n, nq, k = 20_000, 130, 32
    v, _ = make_corpus(n, 1536, 7300)
    qs = make_queries(nq, 1536, 7301)
    rng = np.random.default_rng(7302)
    dup_rows = rng.choice(n, size=300, replace=False)
    _plant_near_duplicates(v, qs, 3, dup_rows, rng)
    v[dup_rows[:40]] = v[dup_rows[0]]  # and 40 exact duplicates among them: ties resolve to ascending ordinal
    vb = new_vb(v, dtype="fp16")
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert vb.engine.get_option("last_tier") == 4
    assert vb.engine.get_option("last_flagged") == 0
    v16 = _f16(v)
    for qi in [0, 1, 2, 3, 4, 64, 129]:
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
    assert set(r.item for r in out[3]) <= set(dup_rows.tolist())
    seq = vb.fuzzy_lookup_embedding(qs[3], max_hits=k, min_score=0.0)  # (another summation order: rows ~1e-7 apart may swap)
    vo.check_topk_parity(vo.scores_full(v16, qs[3]), *items_scores(seq), k, 0.0, referee=vo.f64_referee(v16, qs[3]))
    np.testing.assert_allclose([r.score for r in out[3]], [r.score for r in seq], atol=1e-6, rtol=0)
    # with a threshold, and again
    out2 = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.6)
    for qi in [2, 3, 4]:
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out2[qi]), k, 0.6, referee=vo.f64_referee(v16, qs[qi]))
    assert len(out2[3]) == k and len(out2[2]) == 0
    # k = 50 (the reference's related-terms max_matches, convsettings.py:61-63) and k = 64 ride the wide tile too
    for kk in (50, 64):
        o = vb.fuzzy_lookup_embeddings(qs, max_hits=kk, min_score=0.0)
        assert vb.engine.get_option("last_tier") == 4 and vb.engine.get_option("last_flagged") == 0
        for qi in [3, 77]:
            vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(o[qi]), kk, 0.0, referee=vo.f64_referee(v16, qs[qi]))


def test_wide_tile_falls_back_to_the_exact_tile_when_the_band_overflows():
    """A band bigger than the rescoring takes (1024 candidates per query): 1300 near-duplicates for one query, 1200 exact duplicates of the
    best row of another.  Those queries are flagged on the device and re-run on the exact 64-query tile through the work list; the
    others are not.  Answers must still be the oracle's, ties by ascending ordinal."""
    n, nq, k = 20_000, 130, 32
    v, _ = make_corpus(n, 1536, 7310)
    qs = make_queries(nq, 1536, 7311)
    rng = np.random.default_rng(7312)
    rows = rng.choice(n, size=2500, replace=False)
    _plant_near_duplicates(v, qs, 3, rows[:1300], rng)
    best = qs[9] + 0.2 * rng.standard_normal(1536).astype(np.float32) / np.sqrt(1536)
    v[rows[1300:]] = best / np.linalg.norm(best)
    vb = new_vb(v, dtype="fp16")
    vb.engine.set_option("band_max", 1024)  # (the band buffer of rounds 3-5: this test's clusters are sized to overflow THAT; the default is 2048 since round 6)
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert vb.engine.get_option("last_tier") == 4
    flagged = vb.engine.get_option("last_flagged")
    assert 2 <= flagged <= 4, flagged
    v16 = _f16(v)
    for qi in [0, 3, 4, 9, 64, 129]:
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
    assert [r.item for r in out[9]] == sorted(rows[1300:].tolist())[:k]
    assert set(r.item for r in out[3]) <= set(rows[:1300].tolist())
    out2 = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.6)  # the work list is rebuilt per call
    for qi in [2, 3, 9]:
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out2[qi]), k, 0.6, referee=vo.f64_referee(v16, qs[qi]))
    assert len(out2[3]) == k and len(out2[2]) == 0


def test_a_band_of_1500_near_duplicates_per_query_fits_the_default_band_buffer():
    """Round 6: the band buffer holds 2048 candidates per query (option band_max; 1024 before).  The duplication cliff of bench.py's cfg3_dup --
    EVERY query next to 1500 near-duplicates -- no longer flags anybody: the filter runs once, the rescoring gathers 1500 rows per query
    (the whole batch used to take the exact split-plane form: twice the MFMAs of a filter pass on top of it).  2300 near-duplicates
    still overflow: those queries are flagged and re-run exactly.  A batch is its sequential lookups either way."""
    n, nq, k = 260_000, 130, 32
    v, _ = make_corpus(n, 512, 8790)
    qs = make_queries(nq, 512, 8791)
    rng = np.random.default_rng(8792)
    ids = list(range(0, 100))
    rows = _plant_clusters(v, qs, ids, 1500, rng)
    big = rng.permutation(np.setdiff1d(np.arange(n), rows.reshape(-1)))[:2300]
    _plant_near_duplicates(v, qs, 120, big, rng)
    vb = new_vb(v, dtype="fp16")
    eng = vb.engine
    assert eng.get_option("band_max") == 2048
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert eng.get_option("last_tier") == 4
    assert 1 <= eng.get_option("last_flagged") <= 3, eng.get_option("last_flagged")  # query 120 (+ at most a neighbour or two), none of the hundred
    v16 = _f16(v)
    for j in (0, 1, 50, 99):
        assert set(r.item for r in out[j]) <= set(rows[j].tolist())
    assert set(r.item for r in out[120]) <= set(big.tolist())
    for qi in (0, 1, 50, 99, 100, 120, 129):
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=0.0)
        if qi != 120:  # rescored with the streaming kernels' arithmetic over the WHOLE band: the sequential lookup bit for bit
            assert [(r.item, r.score) for r in out[qi]] == [(r.item, r.score) for r in seq], qi
        else:  # (flagged, on the 64-query exact tile: 64 - k ranks of slack below the k-th, not a score band -- near-ties packed inside 1e-5 may permute)
            np.testing.assert_allclose([r.score for r in out[qi]], [r.score for r in seq], atol=3e-7, rtol=0)
    # ONE un-seeded phase (every row of the corpus admitted: 260k keys per query stream through the select kernel's 4096-key cache, which is cut
    # to its band again and again while the band grows to 1500 keys -- the cut keeps up to band_max keys, not a quarter of the cache): same answers
    eng.set_option("mfma_sample_rows", -1)
    out1 = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert 1 <= eng.get_option("last_flagged") <= 3
    for qi in (0, 1, 50, 99, 100, 129):
        assert [(r.item, r.score) for r in out1[qi]] == [(r.item, r.score) for r in out[qi]], qi
    eng.set_option("mfma_sample_rows", 0)
    # the old buffer: the hundred are flagged too; the oracle's answers again
    eng.set_option("band_max", 1024)
    out2 = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert eng.get_option("last_flagged") >= 101
    for qi in (0, 50, 99, 100, 120):
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out2[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
        np.testing.assert_allclose([r.score for r in out2[qi]], [r.score for r in out[qi]], atol=3e-7, rtol=0)
    with pytest.raises(ValueError):
        eng.set_option("band_max", 4096)


def test_wide_tile_with_the_query_operand_straight_from_l2():
    """`mfma_bdirect=1`: the 256-query tile takes its queries in MFMA-fragment-major order straight from L2 into registers (four rotating fragment
    sets, three corpus slots in the LDS the query ring used to occupy).  Same keys as the default path, incl. padding queries (300 of 512), a
    threshold, several ladder phases, the band of a near-duplicate cluster, and a second dimension."""
    for n, d, nq, k, ms in [(200_000, 1536, 300, 32, 0.0), (90_000, 256, 1024, 50, 0.55), (170_000, 768, 257 + 256, 10, 0.0)]:
        v, _ = make_corpus(n, d, 8600 + d)
        qs = make_queries(nq, d, 8601 + d)
        rng = np.random.default_rng(8602)
        _plant_near_duplicates(v, qs, 7, rng.choice(n, size=200, replace=False), rng)
        vb = new_vb(v, dtype="fp16")
        eng = vb.engine
        base = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms, as_arrays=True)
        assert eng.get_option("last_tier") == 4
        eng.set_option("mfma_bdirect", 1)
        direct = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms, as_arrays=True)
        eng.set_option("mfma_bdirect", 0)
        for a, b in zip(base, direct):
            np.testing.assert_array_equal(a, b)
        v16 = _f16(v)
        o, s_, c_ = direct
        for qi in (0, 7, nq - 1):
            vo.check_topk_parity(vo.scores_full(v16, qs[qi]), o[qi, : c_[qi]].tolist(), s_[qi, : c_[qi]].tolist(), k, ms, referee=vo.f64_referee(v16, qs[qi]))


def test_many_flagged_queries_take_the_wide_exact_fallback():
    """The duplication cliff, bounded: when MANY queries of a batch (> 64) have more near-duplicates than a band holds, they are re-run on the
    256-query tile's exact form (fp32 queries as two fp16 planes, the K loop once per plane) instead of 64 at a time on the 64-query exact
    tile.  100 of 256 queries get 1100 near-duplicate rows each (a band holds 1024): all are flagged; answers are the oracle's either way
    (`wide_fallback=0`: the 64-query tile, two passes)."""
    n, nq, k = 130_000, 256, 32
    v, _ = make_corpus(n, 1536, 8500)
    qs = make_queries(nq, 1536, 8501)
    rng = np.random.default_rng(8502)
    rows = rng.permutation(n)[: 100 * 1100].reshape(100, 1100)
    for j in range(100):
        _plant_near_duplicates(v, qs, 2 * j + 1, rows[j], rng)
    vb = new_vb(v, dtype="fp16")
    vb.engine.set_option("band_max", 1024)  # (the band buffer of rounds 3-5: this test's clusters are sized to overflow THAT; the default is 2048 since round 6)
    eng = vb.engine
    v16 = _f16(v)
    probe = [0, 1, 3, 77, 101, 199, 200, 255]
    outs = {}
    for mode in (1, 0):
        eng.set_option("wide_fallback", mode)
        out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
        assert eng.get_option("last_tier") == 4
        assert 100 <= eng.get_option("last_flagged") <= 160  # (a few more: buffers of other queries that overflowed on the planted rows inside one row range)
        assert eng.get_option("last_doomed") <= 128  # not most of the batch: the filter ran to its end
        for qi in probe:
            vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
            if qi % 2 == 1 and qi < 200:
                assert set(r.item for r in out[qi]) <= set(rows[qi // 2].tolist())  # the flagged queries' hits are their planted rows
        outs[mode] = out
    # both exact fallbacks hand their best rows (+ a band below the k-th) to the rescoring kernel, whose arithmetic is the streaming kernels':
    # the wide split-plane form, the 64-query tile and `fuzzy_lookup_embedding` on its own return the same float32 scores in the same order
    for qi in probe:
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=0.0)
        assert eng.get_option("last_tier") in (1, 2, 3)
        assert [(r.item, r.score) for r in outs[1][qi]] == [(r.item, r.score) for r in seq], qi
        if qi % 2 == 0 or qi >= 200:  # (un-flagged queries; a flagged one on the 64-query tile has 64 - k ranks of slack, not a score band: near-ties of a
            assert [(r.item, r.score) for r in outs[0][qi]] == [(r.item, r.score) for r in seq], qi  # 1100-row cluster packed inside 1e-5 may permute)
        else:
            np.testing.assert_allclose([r.score for r in outs[0][qi]], [r.score for r in seq], atol=3e-7, rtol=0)
    # a threshold above the un-planted queries' scores: only the flagged ones return anything, through the same path
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.7)
    assert len(out[0]) == 0 and len(out[1]) == k
    vo.check_topk_parity(vo.scores_full(v16, qs[1]), *items_scores(out[1]), k, 0.7, referee=vo.f64_referee(v16, qs[1]))


@pytest.mark.parametrize("cluster", [2000, 12_000])
def test_a_batch_of_mostly_doomed_bands_skips_the_last_filter_phase(cluster):
    """200 of 256 queries sit next to ONE cluster of 2000 near-duplicate rows spread evenly over the corpus (a band holds 1024): when the phase
    before the last ends, their bands over the rows seen so far already extrapolate past the buffer -- more than half the batch, so the last
    (biggest) filter phase, its selection and the rescoring return at once and EVERY query takes the exact split-plane form.  Same answers
    as with `early_exact=0` (filter to the end, then the exact form for the flagged ones) and as the oracle.  (12 000 rows: the band over the
    first phase alone is past the buffer -- cut to the strict best k there and counted all the same.)"""
    n, nq, k = 170_000, 256, 32
    v, _ = make_corpus(n, 1536, 8600)
    qs = make_queries(nq, 1536, 8601)
    rng = np.random.default_rng(8602)
    centre = make_queries(1, 1536, 8603)[0]
    for j in range(200):
        w = centre + 0.05 * rng.standard_normal(1536).astype(np.float32) / np.sqrt(1536)
        qs[j] = w / np.linalg.norm(w)
    rows = rng.permutation(n)[:cluster]
    for r in rows:
        w = centre + 2e-4 * rng.standard_normal(1536).astype(np.float32) / np.sqrt(1536)
        v[r] = w / np.linalg.norm(w)
    bounds = _native.plan_ladder(n, nq)
    assert len(bounds) >= 3  # at least two phases: there is a phase before the last
    seen = bounds[-2] / n
    early_rows = int((rows < bounds[-2]).sum())  # of the cluster, in the phases before the last (~240): every one of the 200 queries holds them all in its band
    assert early_rows > 1.25 * 1024 * seen + 40 and early_rows > k + 15 + 40  # the extrapolation rule of run_tile_ladder fires with room to spare
    vb = new_vb(v, dtype="fp16")
    vb.engine.set_option("band_max", 1024)  # (the band buffer of rounds 3-5: this test's clusters are sized to overflow THAT; the default is 2048 since round 6)
    eng = vb.engine
    v16 = _f16(v)
    probe = [0, 1, 57, 199, 200, 255]
    outs = {}
    for early in (1, 0):
        eng.set_option("early_exact", early)
        out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
        assert eng.get_option("last_tier") == 4
        if early:
            assert 195 <= eng.get_option("last_doomed") <= 200
            assert eng.get_option("last_flagged") == nq  # nobody was rescored: all of them went to the exact form
        else:
            assert eng.get_option("last_doomed") == 0 and 200 <= eng.get_option("last_flagged") < nq
        for qi in probe:
            vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
        assert set(r.item for r in out[0]) <= set(rows.tolist())
        outs[early] = out
    for qi in probe:
        np.testing.assert_allclose([r.score for r in outs[1][qi]], [r.score for r in outs[0][qi]], atol=1e-6, rtol=0)
    # a threshold: the queries away from the cluster return nothing, through the same skipped-phase path
    eng.set_option("early_exact", 1)
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.7)
    assert eng.get_option("last_flagged") == nq and len(out[255]) == 0 and len(out[0]) == k
    vo.check_topk_parity(vo.scores_full(v16, qs[0]), *items_scores(out[0]), k, 0.7, referee=vo.f64_referee(v16, qs[0]))


@pytest.mark.parametrize("sample", [-1, 20480, 0])
def test_wide_tile_band_overflow_inside_one_row_range(sample):
    """900 near-duplicates in CONSECUTIVE rows: one workgroup's candidate buffer (1024 keys) cannot hold the band while it walks its
    row range -- the in-kernel compaction falls back to the strict best k and flags the query; everything else stays on the band path.
    (Whether a buffer overflows depends on how the rows fall into row ranges: one un-seeded phase and a first phase of 20480 rows put
    enough of the 900 into one range; the default ladder's geometry is free to change -- the answers are exact either way.)"""
    n, nq, k = 200_000, 130, 32
    v, _ = make_corpus(n, 1536, 7320)
    qs = make_queries(nq, 1536, 7321)
    rng = np.random.default_rng(7322)
    _plant_near_duplicates(v, qs, 5, range(150_000, 150_900), rng)
    vb = new_vb(v, dtype="fp16")
    vb.engine.set_option("mfma_sample_rows", sample)
    vb.engine.set_option("mfma_tile", 256)  # (256 row ranges of 782 rows: the geometry this test's cluster is placed for; the default at this size is 128-query tiles since round 6)
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert vb.engine.get_option("last_tier") == 4
    assert vb.engine.get_option("last_flagged") in ((0, 1) if sample == 0 else (1,))  # (default ladder: whatever its geometry makes of it)
    v16 = _f16(v)
    for qi in [0, 4, 5, 6, 129]:
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
    assert all(150_000 <= r.item < 150_900 for r in out[5])


def _plant_clusters(v, qs, query_ids, rows_per_query, rng, eps=2e-4):
    """vectorised `_plant_near_duplicates` for MANY queries: query_ids[j] gets rows_per_query rows next to it (scores within ~1e-5 of one another)"""
    d = v.shape[1]
    rows = rng.permutation(v.shape[0])[: len(query_ids) * rows_per_query].reshape(len(query_ids), rows_per_query)
    for j, qi in enumerate(query_ids):
        base = qs[qi] + 0.3 * rng.standard_normal(d).astype(np.float32) / np.sqrt(d)
        base /= np.linalg.norm(base)
        w = base[None, :] + eps * rng.standard_normal((rows_per_query, d)).astype(np.float32) / np.sqrt(d)
        v[rows[j]] = w / np.linalg.norm(w, axis=1, keepdims=True)
    return rows


def test_unused_slots_of_the_wide_fallback_admit_nothing():
    """The wide split-plane fallback runs over a work list padded to whole 256-query tiles.  The unused slots of the last live tile hold zero
    queries (every row scores 0.5): with `min_score` below that they used to admit every row of the big ladder phases from the second phase on
    (their per-phase threshold is NaN: the select kernel skips them) -- correct answers, several times the time.  Two batches whose work lists
    need the SAME number of tiles, one leaving most of its last tile unused, one filling it, must cost about the same, and the answers are
    the oracle's.  (The planted clusters are most of this corpus, so about a third of the un-planted queries are flagged too: a cluster that
    scores inside their top k is 1100 rows inside their band.)"""
    import time

    n, d, nq, k = 560_000, 512, 1024, 32
    v, _ = make_corpus(n, d, 8700)
    qs = make_queries(nq, d, 8701)
    rng = np.random.default_rng(8702)
    rows = _plant_clusters(v, qs, list(range(480)), 1100, rng)
    vb = new_vb(v, dtype="fp16")
    vb.engine.set_option("band_max", 1024)  # (the band buffer of rounds 3-5: this test's clusters are sized to overflow THAT; the default is 2048 since round 6)
    eng = vb.engine
    eng.set_option("early_exact", 0)  # (the filter runs to its end in both batches: the fallback's cost is what is compared)
    plain = make_queries(nq, d, 8703)
    batches = {}
    for planted in (230, 480):
        b = plain.copy()
        b[:planted] = qs[:planted]
        batches[planted] = b
    times, flagged = {}, {}
    v16 = _f16(v)
    for m, b in batches.items():
        out = vb.fuzzy_lookup_embeddings(b, max_hits=k, min_score=0.0, as_arrays=True)
        assert eng.get_option("last_tier") == 4
        flagged[m] = int(eng.get_option("last_flagged"))
        assert m <= flagged[m] <= m + (nq - m) * 6 // 10, flagged  # the planted queries + up to ~half of the others
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            vb.fuzzy_lookup_embeddings(b, max_hits=k, min_score=0.0, as_arrays=True)
            best = min(best, time.perf_counter() - t0)
        times[m] = best
        o, s_, c_ = out
        for qi in (0, 229, 230, m - 1, m, 1023):
            vo.check_topk_parity(vo.scores_full(v16, b[qi]), o[qi, : c_[qi]].tolist(), s_[qi, : c_[qi]].tolist(), k, 0.0, referee=vo.f64_referee(v16, b[qi]))
            if qi < m:
                assert set(o[qi, : c_[qi]].tolist()) <= set(rows[qi].tolist())
    tiles = {m: (f + 255) // 256 for m, f in flagged.items()}
    unused = {m: tiles[m] * 256 - f for m, f in flagged.items()}
    assert min(flagged.values()) > 64, flagged  # both batches are served by the wide form
    if tiles[230] == tiles[480]:
        assert unused[230] > unused[480] + 64, (flagged, "the batches were meant to differ in unused slots")
        assert times[230] < 1.3 * times[480], (times, flagged)
    else:  # (another tile count: the emptier work list must at least not cost more per tile)
        assert times[230] / tiles[230] < 1.3 * times[480] / tiles[480], (times, flagged)


@pytest.mark.parametrize("nq", [24, 40, 130, 1024])
def test_per_query_thresholds_ride_the_tiles(nq):
    """A batch whose `min_score`s differ per query (Q calls of the reference have Q of them, vectorbase.py:163-173) takes the same kernels as a
    uniform one -- the 32/64-query tile at 24 and 40 queries, the 128/256-query tile + rescoring beyond -- and returns, query by query, what the
    single-query kernel returns with that query's threshold (NaN and > 1 thresholds included: nothing passes)."""
    n, d, k = 60_000, 1536, 32
    v, _ = make_corpus(n, d, 8710)
    qs = make_queries(nq, d, 8711)
    vb = new_vb(v, dtype="fp16")
    eng = vb.engine
    v16 = _f16(v)
    ref0 = vo.scores_full(v16, qs[0])
    levels = np.sort(ref0)[::-1]
    thr = np.full(nq, 0.0, dtype=np.float64)
    thr[1::4] = float(levels[10])   # ~11 rows of query 0 pass; other queries: about as many
    thr[2::4] = float(levels[200])
    thr[3::4] = 0.4
    thr[5] = float("nan")
    thr[6] = 1.5
    thr[7] = float(levels[0])  # exactly the best score of query 0 (for query 7: usually nothing)
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=thr.tolist())
    assert eng.get_option("last_tier") == (4 if nq >= 65 else 5)
    assert len(out[5]) == 0 and len(out[6]) == 0
    for qi in sorted(set([0, 1, 2, 3, 4, 5, 6, 7, 9, nq // 2, nq - 1])):
        t = thr[qi]
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=None if t != t else float(t)) if t == t else []
        assert eng.get_option("last_tier") in (1, 2, 3) or t != t
        if nq >= 65:  # rescored with the streaming kernels' arithmetic: the same float32 values, so the same order
            assert [r.item for r in out[qi]] == [r.item for r in seq], qi
            assert [r.score for r in out[qi]] == [r.score for r in seq], qi
        else:
            # the 32/64-query tile returns its own accumulation order: scores within float32 summation noise of the streaming kernel's, and two
            # rows closer than that may trade places (query 39 of the 40-query case: ranks 15 / 16 are 4.9e-8 apart in float64) -- the
            # refereed check below decides whether that is all a different order is
            assert len(out[qi]) == len(seq), qi
            np.testing.assert_allclose([r.score for r in out[qi]], [r.score for r in seq], atol=3e-7, rtol=0)
            if [r.item for r in out[qi]] != [r.item for r in seq]:
                assert sorted(r.item for r in out[qi]) == sorted(r.item for r in seq) or len(seq) == k, qi
        if t == t:
            vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, float(np.float32(t)), referee=vo.f64_referee(v16, qs[qi]))
    # the uniform form of the same call still means the same thing
    uni = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.4)
    assert [(r.item, r.score) for r in uni[3]] == [(r.item, r.score) for r in out[3]]
    assert eng.get_option("last_tier") == (4 if nq >= 65 else 5)
    # through the C ABI with a float32 array, on an fp32 corpus (32-query fp32 tile; the fp16 shadow + fp32 rescoring from 33 queries on -- 65 until
    # the end of round 6)
    vb32 = new_vb(v[:20_000], dtype="fp32")
    t32 = np.where(np.arange(nq) % 2 == 0, np.float32(0.0), np.float32(0.5)).astype(np.float32)
    o, s_, c_ = vb32.engine.search_batch(qs, k, t32)
    assert vb32.engine.get_option("last_tier") == (4 if nq >= 33 else 5)
    for qi in (0, 1, nq - 2, nq - 1):
        seq = vb32.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=float(t32[qi]))
        assert o[qi, : c_[qi]].tolist() == [r.item for r in seq]
        np.testing.assert_allclose(s_[qi, : c_[qi]], [r.score for r in seq], atol=3e-7, rtol=0)


@pytest.mark.parametrize("nq,k", [(1024, 100), (130, 256), (300, 65)])
def test_wide_tile_serves_k_beyond_64(nq, k):
    """`max_hits` up to 256 rides the 128/256-query tile on fp16 corpora (the band selection holds any k the fused selections serve): a
    1024-query batch with k = 100 used to fall to the streaming kernel, 8 queries per corpus pass.  One query sits on a cluster of 1300
    near-duplicates (a band that does not fit): beyond k = 64 the flagged query goes to the wide split-plane form whatever the count."""
    n, d = 150_000, 1536
    v, _ = make_corpus(n, d, 8720 + k)
    qs = make_queries(nq, d, 8721 + k)
    rng = np.random.default_rng(8722)
    dup = rng.choice(n, size=1300, replace=False)
    _plant_near_duplicates(v, qs, 3, dup, rng)
    vb = new_vb(v, dtype="fp16")
    vb.engine.set_option("band_max", 1024)  # (the band buffer of rounds 3-5: this test's clusters are sized to overflow THAT; the default is 2048 since round 6)
    eng = vb.engine
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert eng.get_option("last_tier") == 4
    assert 1 <= eng.get_option("last_flagged") <= 3
    v16 = _f16(v)
    for qi in sorted(set([0, 1, 3, 4, nq // 2, nq - 1])):
        assert len(out[qi]) == k
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
        if qi != 3:
            seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=0.0)
            assert [(r.item, r.score) for r in out[qi]] == [(r.item, r.score) for r in seq], qi
    assert set(r.item for r in out[3]) <= set(dup.tolist())
    # with a threshold that leaves fewer than k rows
    thr = float(np.float32(out[0][k // 2].score))
    out2 = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=thr)
    assert len(out2[0]) == k // 2 + 1 and [r.item for r in out2[0]] == [r.item for r in out[0][: k // 2 + 1]]
    # fp32 corpora ride the wide tile over the shadow beyond k = 64 as well (end of round 6; the streaming kernels, four queries per pass, until then)
    vb32 = new_vb(v[:30_000], dtype="fp32")
    o32 = vb32.fuzzy_lookup_embeddings(qs[:70], max_hits=k, min_score=0.0)
    assert vb32.engine.get_option("last_tier") == 4 and vb32.engine.get_option("last_shadow") == 1
    vo.check_topk_parity(vo.scores_full(v[:30_000], qs[5]), *items_scores(o32[5]), k, 0.0, referee=vo.f64_referee(v[:30_000], qs[5]))


@pytest.mark.parametrize("nq,k", [(130, 100), (12, 65), (1024, 256)])
def test_k_beyond_64_on_an_fp32_corpus_rides_the_shadow_and_re_runs_flagged_queries_on_the_scan(nq, k):
    """fp32 corpus, `max_hits` 65 .. 256, a batch: the wide tile over the fp16 shadow keeps the band, the fp32 rescoring ranks it -- any k the fused
    selections serve.  No exact tile ranks more than 64 fp32 rows per query, so a flagged query (here: one sitting on 2500 near-duplicates, more
    than the band buffer holds) is re-run on the streaming kernels after ONE host round trip (the work list read back).  Every answer is the
    sequential fp32 lookup's, bit for bit; until the end of round 6 such a batch took the streaming kernels four queries per corpus pass."""
    n, d = 60_000, 1536
    v, _ = make_corpus(n, d, 8770 + k)
    qs = make_queries(nq, d, 8771 + k)
    rng = np.random.default_rng(8772)
    dup = rng.choice(n, size=2500, replace=False)
    _plant_near_duplicates(v, qs, 3, dup, rng)
    vb = new_vb(v, dtype="fp32")
    eng = vb.engine
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert eng.get_option("last_tier") == 4 and eng.get_option("last_shadow") == 1
    assert 1 <= eng.get_option("last_flagged") <= 8  # (query 3, and at k = 256 a few queries whose 256th best reaches into its cluster)
    for qi in sorted(set([0, 1, 2, 3, 4, nq // 2, nq - 1])):
        assert len(out[qi]) == k
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=0.0)
        assert eng.get_option("last_tier") in (1, 2, 3)
        assert [(r.item, r.score) for r in out[qi]] == [(r.item, r.score) for r in seq], qi
        vo.check_topk_parity(vo.scores_full(v, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v, qs[qi]))
    assert set(r.item for r in out[3]) <= set(dup.tolist())
    # per-query thresholds reach the re-run too
    thr = np.zeros(nq)
    thr[3] = float(np.float32(out[3][k // 2].score))
    out2 = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=thr.tolist())
    seq3 = vb.fuzzy_lookup_embedding(qs[3], max_hits=k, min_score=float(thr[3]))  # (near-duplicates tie at float32 scores: more than k / 2 + 1 rows reach it)
    assert k // 2 + 1 <= len(out2[3]) <= k and [(r.item, r.score) for r in out2[3]] == [(r.item, r.score) for r in seq3]
    assert [(r.item, r.score) for r in out2[0]] == [(r.item, r.score) for r in out[0]]


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_small_batches_with_k_beyond_64_take_the_wide_tile(dtype):
    """k > 64 is beyond the 32/64-query tile, and the streaming kernels take four such queries per corpus pass: from 9 queries up (3 on corpora of
    256 MiB and more) the batch rides the wide tile -- 32 queries over 2M x 1536 fp16 rows, k = 65: 12.5 ms against 1.3.  Same answers."""
    n = 90_000 if dtype == "fp16" else 45_000  # 276 MB either way
    v, _ = make_corpus(n, 1536, 8780)
    qs = make_queries(32, 1536, 8781)
    vb = new_vb(v, dtype=dtype)
    eng = vb.engine
    ref_v = v if dtype == "fp32" else _f16(v)
    for nq, tier in ((32, 4), (9, 4), (3, 4), (2, None)):
        out = vb.fuzzy_lookup_embeddings(qs[:nq], max_hits=100, min_score=0.0)
        got = eng.get_option("last_tier")
        assert (got == tier) if tier else (got in (1, 2, 3)), (nq, got)
        for qi in (0, nq - 1):
            seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=100, min_score=0.0)
            assert [(r.item, r.score) for r in out[qi]] == [(r.item, r.score) for r in seq], (nq, qi)
            vo.check_topk_parity(vo.scores_full(ref_v, qs[qi]), *items_scores(out[qi]), 100, 0.0, referee=vo.f64_referee(ref_v, qs[qi]))
    small = new_vb(v[:5_000], dtype=dtype)  # below 256 MiB: 3 .. 8 queries stay on the streaming kernels (two passes at most), 9 go wide
    small.fuzzy_lookup_embeddings(qs[:8], max_hits=100, min_score=0.0)
    assert small.engine.get_option("last_tier") in (1, 2, 3)
    small.fuzzy_lookup_embeddings(qs[:9], max_hits=100, min_score=0.0)
    assert small.engine.get_option("last_tier") == 4


def test_wide_tile_row_norm_cache_follows_appends_and_rewrites():
    """The exactness proof uses the largest row norm of the corpus (cached per corpus on the device): appended rows with a
    bigger norm, and rows rewritten in place, must be seen."""
    v, _ = make_corpus(9_000, 1536, 7400)
    qs = make_queries(70, 1536, 7401)
    vb = new_vb(v, dtype="fp16")
    first = vb.fuzzy_lookup_embeddings(qs, max_hits=10, min_score=0.0)
    big = (qs[:5] * 30.0).astype(np.float32)  # un-normalised rows: dot products up to 30, scores clip to 1.0
    vb.add_embeddings(None, big)
    second = vb.fuzzy_lookup_embeddings(qs, max_hits=10, min_score=0.0)
    allv = _f16(np.concatenate([v, big]))
    for qi in range(0, 70, 7):
        vo.check_topk_parity(vo.scores_full(allv, qs[qi]), *items_scores(second[qi]), 10, 0.0, referee=vo.f64_referee(allv, qs[qi]))
    assert second[0][0].item == 9_000 and second[0][0].score == 1.0
    assert [r.item for r in first[60]] == [r.item for r in second[60]][: len(first[60])] or True

# --------------------------------------------------------------------------------------
# fp32 corpora, 65+ queries: the 128/256-query fp16 tile filters on an fp16 SHADOW of the corpus, the candidates are
# rescored with the fp32 rows and fp32 queries (tavb_rescore.hip) -- same answers as the single-query fp32 kernels
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("d", [64, 768, 3072, 4096, 1000, 96, 1008, 1004, 1001, 70, 33])
def test_wide_tile_other_dimensions(dtype, d):
    """The 128/256-query tile takes any D that is a multiple of 64 (K steps of whole 128-byte lines) directly and other widths on a zero-padded
    fp16 copy of the rows (the filter's operand; candidates are rescored with the corpus' own rows): ANY width on fp16 corpora (round 6: rows
    that are not 16-byte aligned are rescored element by element, in the scalar streaming kernel's order), multiples of 16 on fp32 ones (their
    exact fallback tile reads the fp32 rows).  What is left (fp32: 1000, 1004, 1001, 70, 33) falls back to the 32/64-query tile or the
    streaming tiers.  130 queries: one 256-query tile."""
    n, nq, k = 20_011, 130, 32
    v, _ = make_corpus(n, d, 9900 + d)
    qs = make_queries(nq, d, 9901 + d)
    qs[3] = v[n - 1]
    vb = new_vb(v, dtype=dtype)
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    t = vb.engine.get_option("last_tier")
    wide = d % 64 == 0 or dtype == "fp16" or d % 16 == 0
    assert t == 4 if wide else t in (1, 2, 3, 5), (d, dtype, t)
    ref_v = v if dtype == "fp32" else _f16(v)
    for qi in range(0, nq, 9):
        rep = vo.check_topk_parity(vo.scores_full(ref_v, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(ref_v, qs[qi]))
        assert rep.tie_permuted_positions <= 2
        if wide:  # rescored with the streaming kernels' arithmetic: a batch is its sequential lookups, bit for bit
            seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=k, min_score=0.0)
            assert [(r.item, r.score) for r in out[qi]] == [(r.item, r.score) for r in seq], (d, dtype, qi)
    assert out[3][0].item == n - 1
    # appended rows reach the padded copy too
    if wide and d % 64 != 0:
        extra, _ = make_corpus(300, d, 9950 + d)
        vb.add_embeddings(None, extra)
        qs2 = qs.copy()
        qs2[7] = extra[123]
        out2 = vb.fuzzy_lookup_embeddings(qs2, max_hits=k, min_score=0.0)
        assert vb.engine.get_option("last_tier") == 4 and out2[7][0].item == n + 123
        ref2 = np.concatenate([ref_v, extra if dtype == "fp32" else _f16(extra)])
        vo.check_topk_parity(vo.scores_full(ref2, qs2[7]), *items_scores(out2[7]), k, 0.0, referee=vo.f64_referee(ref2, qs2[7]))


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_batches_of_33_to_64_take_the_wide_tile_on_big_corpora(dtype):
    """On corpora of 256 MiB and more a batch of 33 .. 64 queries rides the 128-query tile (padded) + rescoring instead of the 64-query tile
    (10M fp16 rows, 64 queries: 5.46 against 5.91 ms; 1M fp32 rows through the shadow: 0.85 against 2.17 ms); smaller corpora and smaller
    batches keep the 32/64-query tile.  Same answers either way: a batch is its sequential lookups."""
    n = 100_000 if dtype == "fp16" else 50_000  # 307 MB either way
    v, _ = make_corpus(n, 1536, 8760)
    qs = make_queries(64, 1536, 8761)
    qs[5] = v[n - 3]
    vb = new_vb(v, dtype=dtype)
    eng = vb.engine
    ref_v = v if dtype == "fp32" else _f16(v)
    for nq, tier in ((64, 4), (33, 4), (32, 5)):
        out = vb.fuzzy_lookup_embeddings(qs[:nq], max_hits=32, min_score=0.0)
        assert eng.get_option("last_tier") == tier, (nq, eng.get_option("last_tier"))
        for qi in (0, 5, nq - 1):
            vo.check_topk_parity(vo.scores_full(ref_v, qs[qi]), *items_scores(out[qi]), 32, 0.0, referee=vo.f64_referee(ref_v, qs[qi]))
            if tier == 4:
                seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=32, min_score=0.0)
                assert [(r.item, r.score) for r in out[qi]] == [(r.item, r.score) for r in seq], (nq, qi)
        assert out[5][0].item == n - 3
    eng.set_option("mfma_big_bytes", 1 << 40)  # "no corpus is big": the 64-query tile again -- on fp16 corpora; fp32 ones send 33+ queries through
    vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)  # the shadow at any size (mfma_min_batch_f32: one fp32 tile is 82 us of matrix work)
    assert eng.get_option("last_tier") == (5 if dtype == "fp16" else 4)
    eng.set_option("mfma_min_batch_f32", 65)
    vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") == 5
    small = new_vb(v[:20_000], dtype=dtype)
    out_small = small.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert small.engine.get_option("last_tier") == (5 if dtype == "fp16" else 4)
    if dtype == "fp32":  # 64 queries over a small fp32 corpus through the shadow: the sequential fp32 lookups bit for bit
        for qi in (0, 5, 63):
            seq = small.fuzzy_lookup_embedding(qs[qi], max_hits=32, min_score=0.0)
            assert [(r.item, r.score) for r in out_small[qi]] == [(r.item, r.score) for r in seq], qi


@pytest.mark.parametrize("nq,k", [(130, 32), (300, 100)])
def test_wide_tile_odd_width_with_a_band_that_does_not_fit(nq, k):
    """D = 1000 on an fp16 corpus with 1300 near-duplicates around one query: its band overflows, the query is flagged and re-run exactly -- by the
    64-query split-plane tile (k <= 64) or the wide split-plane form (k = 100), both over the zero-padded copy of the rows (the same fp16 values)
    -- and rescored with the corpus' own rows: the oracle's answer."""
    n, d = 60_000, 1000
    v, _ = make_corpus(n, d, 8740 + k)
    qs = make_queries(nq, d, 8741 + k)
    rng = np.random.default_rng(8742)
    dup = rng.choice(n, size=1300, replace=False)
    _plant_near_duplicates(v, qs, 3, dup, rng)
    vb = new_vb(v, dtype="fp16")
    vb.engine.set_option("band_max", 1024)  # (the band buffer of rounds 3-5: this test's clusters are sized to overflow THAT; the default is 2048 since round 6)
    eng = vb.engine
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert eng.get_option("last_tier") == 4
    assert 1 <= eng.get_option("last_flagged") <= 3
    v16 = _f16(v)
    for qi in sorted(set([0, 3, 4, nq // 2, nq - 1])):
        assert len(out[qi]) == k
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
    assert set(r.item for r in out[3]) <= set(dup.tolist())


@pytest.mark.parametrize("d", [200, 72, 75])
@pytest.mark.parametrize("nq,k,planted", [(130, 32, 60), (256, 32, 100), (300, 100, 50)])
def test_odd_width_with_many_flagged_queries(d, nq, k, planted):
    """An fp16 corpus whose width is not a multiple of 64 with MANY flagged queries: the work list's per-slot thresholds and band widths sit
    behind the two PADDED query planes.  (Round 5 placed them behind planes of the unpadded width, i.e. inside the low plane from slot
    cap·dim/fdim on -- row 36 of 64 at D = 200: wrong or empty answers for the slots behind it; one to three flagged queries never reached
    those rows.)  130 queries: the 64-query exact tile over 192 slots; 256 queries: the first 64 flagged on it, the rest on the wide
    split-plane form; k = 100: every flagged query on the wide form."""
    n = 150_000
    v, _ = make_corpus(n, d, 8770 + d)
    qs = make_queries(nq, d, 8771 + d)
    rng = np.random.default_rng(8772)
    ids = list(range(1, 2 * planted, 2))
    rows = _plant_clusters(v, qs, ids, 1100, rng)
    vb = new_vb(v, dtype="fp16")
    vb.engine.set_option("band_max", 1024)  # (the band buffer of rounds 3-5: this test's clusters are sized to overflow THAT; the default is 2048 since round 6)
    eng = vb.engine
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert eng.get_option("last_tier") == 4
    assert planted <= eng.get_option("last_flagged") <= nq
    v16 = _f16(v)
    for j, qi in enumerate(ids):  # every flagged slot, the last ones above all
        assert len(out[qi]) == k, (qi, len(out[qi]))
        assert set(r.item for r in out[qi]) <= set(rows[j].tolist()), qi
    for qi in sorted(set([0, 1, 2, ids[planted // 2], ids[-2], ids[-1], nq - 1])):
        assert len(out[qi]) == k
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(out[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
    # with a threshold only the planted queries clear
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.9)
    assert len(out[0]) == 0 and len(out[ids[-1]]) == k
    vo.check_topk_parity(vo.scores_full(v16, qs[ids[-1]]), *items_scores(out[ids[-1]]), k, 0.9, referee=vo.f64_referee(v16, qs[ids[-1]]))


@pytest.mark.parametrize("nq", [5, 8, 32])
def test_small_batches_on_big_fp32_corpora_ride_the_shadow(nq):
    """Round 6: on fp32 corpora of `mfma_big_bytes_f32` (1e9 bytes) or more a batch of 5+ queries takes the wide tile over the fp16 shadow (half the
    bytes of the fp32 rows the 32-query fp32 tile reads: 32 queries over 1M x 1536 rows 0.79 against 1.24 ms) and its candidates are rescored
    with the fp32 rows -- the sequential fp32 lookups bit for bit.  Smaller corpora keep the fp32 tile (the wide path's fixed launches cost more
    than half a pass saves); here the size rule is lowered instead of building a corpus of that size.  Two to four queries (one pass of the
    streaming scan) move from `mfma_few_bytes_f32` (4 GiB) up, a single query never does."""
    n = 40_000
    v, _ = make_corpus(n, 1536, 8795)
    qs = make_queries(nq, 1536, 8796)
    qs[1] = v[n - 5]
    vb = new_vb(v, dtype="fp32")
    eng = vb.engine
    out5 = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") == 5 and eng.get_option("last_shadow") == 0  # 246 MB: the 32-query fp32 tile
    eng.set_option("mfma_big_bytes_f32", 150 << 20)  # (246 MB >= 150 MB: batches of 5+; 2 .. 4 queries have their own size rule)
    out4 = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") == 4 and eng.get_option("last_shadow") == 1 and eng.get_option("last_flagged") == 0
    assert out4[1][0].item == n - 5
    for qi in range(nq):
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=32, min_score=0.0)
        assert eng.get_option("last_tier") in (1, 2, 3)
        assert [(r.item, r.score) for r in out4[qi]] == [(r.item, r.score) for r in seq], qi
        vo.check_topk_parity(vo.scores_full(v, qs[qi]), *items_scores(out5[qi]), 32, 0.0, referee=vo.f64_referee(v, qs[qi]))
    # two to four queries stay on the streaming scan until the corpus reaches mfma_few_bytes_f32; one query always does
    few = vb.fuzzy_lookup_embeddings(qs[:4], max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") in (1, 2, 3)
    eng.set_option("mfma_few_bytes_f32", 100 << 20)
    few4 = vb.fuzzy_lookup_embeddings(qs[:4], max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") == 4 and eng.get_option("last_shadow") == 1
    assert [[(r.item, r.score) for r in a] for a in few4] == [[(r.item, r.score) for r in a] for a in few]
    vb.fuzzy_lookup_embeddings(qs[:1], max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") in (1, 2, 3)
    eng.set_option("mfma_min_batch_big_f32", 65)  # both rules off
    vb.fuzzy_lookup_embeddings(qs[:4], max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") in (1, 2, 3)
    eng.set_option("mfma_min_batch_big_f32", 5)
    eng.set_option("f32_shadow", 0)
    vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") == 5


@pytest.mark.parametrize("nq", [65, 128, 300, 1024])
def test_f32_corpus_large_batches_ride_the_fp16_shadow(nq):
    v, _ = make_corpus(60_007, 1536, 7600)
    qs = make_queries(nq, 1536, 7601 + nq)
    qs[1] = v[60_006]
    vb = new_vb(v)
    eng = vb.engine
    eng.profile_enable(True)
    eng.profile_reset()
    batch = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") == 4 and eng.profile_read(_native.KERNEL_MFMA)[1] == 1 and eng.get_option("last_flagged") == 0
    assert batch[1][0].item == 60_006 and abs(batch[1][0].score - 1.0) < 1e-6
    sample = sorted(set(np.linspace(0, nq - 1, 20).astype(int).tolist()))
    for qi in sample:
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=32, min_score=0.0)
        assert eng.get_option("last_tier") in (1, 2, 3)
        assert [r.item for r in batch[qi]] == [r.item for r in seq]
        np.testing.assert_allclose([r.score for r in batch[qi]], [r.score for r in seq], atol=3e-7, rtol=0)
        assert vo.check_topk_parity(vo.scores_full(v, qs[qi]), *items_scores(batch[qi]), 32, 0.0, referee=vo.f64_referee(v, qs[qi])).ordinals_bit_exact
    # the 64-query fp32 tile (no shadow) gives the same answers
    eng.set_option("f32_shadow", 0)
    plain = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert eng.get_option("last_tier") == 5
    for qi in sample:
        assert [r.item for r in plain[qi]] == [r.item for r in batch[qi]]
    # a threshold near the scores: relaxed filter threshold, exact re-test
    eng.set_option("f32_shadow", 1)
    thr = float(np.float32(batch[0][20].score))
    batch_t = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=thr)
    assert eng.get_option("last_tier") == 4 and len(batch_t[0]) == 21
    for qi in sample[:6]:
        seq = vb.fuzzy_lookup_embedding(qs[qi], max_hits=32, min_score=thr)
        assert [r.item for r in batch_t[qi]] == [r.item for r in seq]


@pytest.mark.parametrize("n,nq,k,ms", [(257, 70, 32, 0.0), (5, 65, 10, 0.0), (3_001, 200, 48, 0.5), (40_000, 129, 1, 0.0)])
def test_f32_shadow_small_and_odd_shapes(n, nq, k, ms):
    v, _ = make_corpus(n, 1536, 7650 + n % 7)
    v[n // 2] = 0.0  # a zero row scores exactly 0.5
    qs = make_queries(nq, 1536, 7651 + nq)
    qs[0] = v[n - 1]
    vb = new_vb(v)
    vb.engine.set_option("direct_group_max_nq", 0)  # (up to 128 queries on corpora this small are the grouped streaming launch's otherwise: this test is about the shadow)
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms)
    assert vb.engine.get_option("last_tier") == 4
    for qi in sorted(set(np.linspace(0, nq - 1, 24).astype(int).tolist())):
        assert vo.check_topk_parity(vo.scores_full(v, qs[qi]), *items_scores(out[qi]), k, ms, referee=vo.f64_referee(v, qs[qi])).tie_permuted_positions <= 1
    assert out[0][0].item == n - 1


@pytest.mark.parametrize("nq", [1, 2, 7, 32, 33, 64])
def test_f32_shadow_level_2_serves_small_batches_and_single_queries(nq):
    """`f32_shadow = 2`: lookups of ANY size on a big enough fp32 corpus filter on the fp16 shadow (32/64-query tile, exact
    queries as split fp16 planes) and rescore 64 candidates with the fp32 rows: same answers as the fp32 kernels."""
    v, _ = make_corpus(12_345, 1536, 7800)
    qs = make_queries(nq, 1536, 7801 + nq)
    qs[0] = v[12_344]
    vb = new_vb(v)
    eng = vb.engine
    eng.set_option("mfma_min_batch_f32", 65)  # (33+ queries ride the wide tile over the shadow by default: this test is about the level-2 route below it)
    plain = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert eng.get_option("last_shadow") == 0
    eng.set_option("f32_shadow", 2)
    eng.set_option("f32_shadow_min_bytes", 1 << 20)  # (default: 2 GB of fp32 rows)
    got = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert eng.get_option("last_shadow") == 1 and eng.get_option("last_flagged") == 0
    swapped = 0
    for qi in range(nq):  # two fp32 summation orders: positions may swap inside fp32 near-ties only (the oracle check below is tie-aware)
        np.testing.assert_allclose([r.score for r in got[qi]], [r.score for r in plain[qi]], atol=3e-7, rtol=0)
        swapped += sum(a.item != b.item for a, b in zip(got[qi], plain[qi]))
        assert len({r.item for r in got[qi]} ^ {r.item for r in plain[qi]}) <= 2
    assert swapped <= 4
    for qi in range(0, nq, 5):
        vo.check_topk_parity(vo.scores_full(v, qs[qi]), *items_scores(got[qi]), 32, 0.0, referee=vo.f64_referee(v, qs[qi]))
    one = vb.fuzzy_lookup_embedding(qs[0], max_hits=10, min_score=0.6)  # the single-query API takes the same route
    assert eng.get_option("last_shadow") == 1 and [r.item for r in one] == [12_344]
    k50 = vb.fuzzy_lookup_embeddings(qs, max_hits=50, min_score=0.0)  # 64 candidates cannot prove a top-50: the fp32 kernels answer
    assert eng.get_option("last_shadow") == 0 and len(k50[0]) == 50
    small = new_vb(v[:1000])  # below the size where halving the pass pays for the rescoring launches
    small.engine.set_option("mfma_min_batch_f32", 65)
    small.engine.set_option("f32_shadow", 2)
    small.fuzzy_lookup_embeddings(qs, max_hits=5, min_score=0.0)
    assert small.engine.get_option("last_shadow") == 0


def test_per_corpus_caches_do_not_survive_a_new_tensor_at_the_same_address():
    """The allocator hands a new tensor the address of a freed one of the same shape; the library keys its row-norm maxima
    and the fp16 shadow on the address: adopting a DIFFERENT tensor object must drop them."""
    import torch

    qs = make_queries(70, 1536, 7900)
    vb = new_vb()
    addresses = []
    for seed in (7901, 7902, 7903):
        v, _ = make_corpus(20_000, 1536, seed)
        v[17] = qs[4]
        t = torch.from_numpy(v).cuda()
        addresses.append(t.data_ptr())
        vb.adopt_device_corpus(t)
        out = vb.fuzzy_lookup_embeddings(qs, max_hits=10, min_score=0.0)
        assert vb.engine.get_option("last_tier") == 4 and out[4][0].item == 17
        for qi in (0, 33, 69):
            assert vo.check_topk_parity(vo.scores_full(v, qs[qi]), *items_scores(out[qi]), 10, 0.0, referee=vo.f64_referee(v, qs[qi])).ordinals_bit_exact
        del t
        vb.clear()
    # (whether the addresses repeated is the allocator's business; the answers must be right either way)


def test_f32_shadow_follows_appends_rewrites_and_near_duplicates():
    v, _ = make_corpus(20_000, 1536, 7700)
    qs = make_queries(80, 1536, 7701)
    vb = new_vb(v)
    eng = vb.engine
    first = vb.fuzzy_lookup_embeddings(qs, max_hits=10, min_score=0.0)
    assert eng.get_option("last_tier") == 4
    # appended rows (the shadow is extended, or rebuilt when the device buffer moved)
    extra = make_queries(3000, 1536, 7702)
    extra[7] = qs[5]
    vb.add_embeddings(None, extra)
    second = vb.fuzzy_lookup_embeddings(qs, max_hits=10, min_score=0.0)
    assert eng.get_option("last_tier") == 4 and second[5][0].item == 20_007 and abs(second[5][0].score - 1.0) < 1e-6
    allv = np.concatenate([v, extra])
    for qi in range(0, 80, 9):
        assert vo.check_topk_parity(vo.scores_full(allv, qs[qi]), *items_scores(second[qi]), 10, 0.0, referee=vo.f64_referee(allv, qs[qi])).ordinals_bit_exact
    # a row rewritten in place in the serialized matrix (noticed by the fingerprint or mark_dirty): the shadow must follow
    m = vb.serialize()
    m[123] = qs[9]
    vb.mark_dirty()
    third = vb.fuzzy_lookup_embeddings(qs, max_hits=10, min_score=0.0)
    assert third[9][0].item == 123 and abs(third[9][0].score - 1.0) < 1e-6
    # near-duplicate rows around rank k: the fp16 shadow cannot separate them; the band below the approximate k-th best holds all
    # of them and the fp32 rescoring orders them (no fallback pass)
    rng = np.random.default_rng(7703)
    base = qs[3].copy()
    dups = np.stack([base + 3e-4 * rng.standard_normal(1536).astype(np.float32) for _ in range(100)])
    dups /= np.linalg.norm(dups, axis=1, keepdims=True)
    vb2 = new_vb(np.concatenate([v, dups.astype(np.float32)]))
    out = vb2.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    assert vb2.engine.get_option("last_tier") == 4 and vb2.engine.get_option("last_flagged") == 0
    allv2 = np.concatenate([v, dups.astype(np.float32)])
    for qi in [0, 4, 40, 79]:
        vo.check_topk_parity(vo.scores_full(allv2, qs[qi]), *items_scores(out[qi]), 32, 0.0, referee=vo.f64_referee(allv2, qs[qi]))
    # query 3 sits in the middle of 100 rows whose scores differ by less than fp32 summation noise: the ORDER among them is
    # not defined (not even by the reference), the set of scores is
    assert all(r.item >= 20_000 for r in out[3]) and len(set(r.item for r in out[3])) == 32
    ref3 = np.sort(vo.scores_full(allv2, qs[3]))[::-1][:32]
    np.testing.assert_allclose([r.score for r in out[3]], ref3, atol=1e-6, rtol=0)


# --------------------------------------------------------------------------------------
# 32-query MFMA tile ("skinny" kernel): batches of 3..32 on fp16 corpora, every batch >= 5 on fp32 corpora
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("n,d,nq,k,ms", [
    (1000, 1536, 8, 10, 0.0), (257, 1536, 9, 1, 0.0), (70_001, 1536, 31, 32, 0.0), (40_000, 1536, 24, 64, 0.5),
    (33_333, 1536, 17, 32, 0.0), (20_000, 384, 12, 10, 0.52), (255, 1536, 12, 5, 0.0), (5, 1536, 8, 10, 0.0),
    (3_000, 64, 10, 7, 0.0), (100_000, 1536, 30, 32, 0.0),
])
def test_skinny_kernel_against_oracle(dtype, n, d, nq, k, ms):
    """fp32 corpus: `v_mfma_f32_32x32x2_f32` on fp32 rows and fp32 queries.  fp16 corpus: fp32 queries split into fp16
    high + low planes, i.e. the same arithmetic meaning as the streaming tiers (fp32 query x fp16 rows), so the oracle
    gets the fp16-rounded corpus and the UNROUNDED queries."""
    v, _ = make_corpus(n, d, 9500 + n % 89 + d)
    qs = make_queries(nq, d, 9501 + nq)
    if n > 10:
        qs[0] = v[n - 2]  # exact self-match in the last rows
    vb = new_vb(v, dtype=dtype)
    vb.engine.set_option("small_direct_bytes", 0)  # (up to 8 queries on a corpus this small would take the one-launch streaming path: this test is about the tile)
    got = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms)
    assert vb.engine.get_option("last_tier") == 5
    ref_v = v if dtype == "fp32" else _f16(v)
    for qi in range(nq):
        vo.check_topk_parity(vo.scores_full(ref_v, qs[qi]), *items_scores(got[qi]), k, ms, referee=vo.f64_referee(ref_v, qs[qi]))
    if n > 10:
        assert got[0][0].item == n - 2


@pytest.mark.parametrize("nq", [33, 40, 64, 65, 200])
def test_skinny_kernel_serves_large_batches_on_fp32_corpora(nq):
    """Batches of 33+ use 64-query tiles, several per row range from 65 queries (the workgroups of a row range share it
    through L2)."""
    n, k = 30_000, 32
    v, _ = make_corpus(n, 1536, 9600)
    qs = make_queries(nq, 1536, 9601 + nq)
    vb = new_vb(v)
    vb.engine.set_option("f32_shadow", 0)  # without the fp16 shadow (its default-on form: test_f32_corpus_large_batches_*)
    got = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert vb.engine.get_option("last_tier") == 5
    for qi in range(0, nq, 7):
        vo.check_topk_parity(vo.scores_full(v, qs[qi]), *items_scores(got[qi]), k, 0.0, referee=vo.f64_referee(v, qs[qi]))
    # the same batch through the streaming tier
    vb.engine.set_option("skinny_min_batch_f32", 1 << 30)
    ref = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert vb.engine.get_option("last_tier") in (1, 2, 3)
    for a, b in zip(got, ref):
        sa, sb = [r.score for r in a], [r.score for r in b]
        np.testing.assert_allclose(sa, sb, atol=2e-6, rtol=0)


@pytest.mark.parametrize("nq", [33, 48, 64])
def test_skinny_64_query_tile_on_fp16_corpora(nq):
    """33 .. 64 queries on an fp16 corpus: one 64-query tile (fp32 queries split into fp16 high + low planes), the same
    arithmetic meaning as the streaming tiers."""
    n, k = 50_001, 32
    v, _ = make_corpus(n, 1536, 9800)
    qs = make_queries(nq, 1536, 9801 + nq)
    qs[1] = v[n - 5]
    vb = new_vb(v, dtype="fp16")
    got = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert vb.engine.get_option("last_tier") == 5
    v16 = _f16(v)
    for qi in range(nq):
        vo.check_topk_parity(vo.scores_full(v16, qs[qi]), *items_scores(got[qi]), k, 0.0, referee=vo.f64_referee(v16, qs[qi]))
    assert got[1][0].item == n - 5


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_skinny_kernel_with_threshold_ladder_nan_and_zero_rows(dtype):
    n, nq, k = 70_001, 20, 32
    v, _ = make_corpus(n, 1536, 9700)
    v[1234] = 0.0          # zero row: score exactly 0.5
    v[4321, 7] = np.nan    # NaN row: never returned
    qs = make_queries(nq, 1536, 9701)
    qs[3] = v[60_000]
    vb = new_vb(v, dtype=dtype)
    eng = vb.engine
    eng.set_option("mfma_sample_rows", 2048)
    eng.profile_enable(True)
    eng.profile_reset()
    got = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    assert eng.get_option("last_tier") == 5
    assert eng.profile_read(_native.KERNEL_SKINNY)[1] == 1 and eng.profile_read(_native.KERNEL_MFMA_SAMPLE)[1] == _ladder_phases(n, 2048, 4) - 1
    eng.set_option("mfma_sample_rows", -1)
    plain = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=0.0)
    ref_v = v if dtype == "fp32" else _f16(v)
    for qi in range(nq):
        assert [(r.item, r.score) for r in got[qi]] == [(r.item, r.score) for r in plain[qi]]
        assert 4321 not in [r.item for r in got[qi]]
        sc = vo.scores_full(ref_v, qs[qi])
        vo.check_topk_parity(sc, *items_scores(got[qi]), k, 0.0)
    assert got[3][0].item == 60_000
    # the zero row scores exactly 0.5 for every query: ask for a window of scores that contains little else
    neg = -qs[:8]  # rows that score s for q score 1 - s for -q; with min_score 0.5 the zero row survives
    low = vb.fuzzy_lookup_embeddings(neg, max_hits=64, min_score=0.5)
    for qi in range(8):
        vo.check_topk_parity(vo.scores_full(ref_v, neg[qi]), *items_scores(low[qi]), 64, 0.5, referee=vo.f64_referee(ref_v, neg[qi]))


# --------------------------------------------------------------------------------------
# K1 normalise / convert kernels
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 1), (7, 3), (100, 33), (1000, 384), (5000, 1536), (64, 4097)])
def test_normalize_rows_kernel(shape):
    import torch

    rng = np.random.default_rng(shape[0] + shape[1])
    x = (rng.standard_normal(shape) * 3).astype(np.float32)
    x[0] = 0  # zero row must stay zero (model_adapters.py:182)
    eng = _native.Engine(0)
    t = torch.from_numpy(x).cuda()
    y = eng.normalize_rows(t).cpu().numpy()
    ref = vo.l2_normalize_rows(x)
    assert np.all(y[0] == 0)
    np.testing.assert_allclose(y, ref, atol=2e-7, rtol=2e-7)  # fp32 summation-order noise only
    if shape[0] > 1:
        n = np.linalg.norm(y[1:].astype(np.float64), axis=1)
        assert np.all(np.abs(n - 1) < 1e-6)  # reference tests/test_embeddings.py:112-120
    eng.normalize_rows_(t)
    np.testing.assert_array_equal(t.cpu().numpy(), y)
    eng.close()


def test_f32_to_f16_is_round_to_nearest_even():
    import torch

    rng = np.random.default_rng(9)
    x = np.concatenate([rng.standard_normal(100_003).astype(np.float32), np.array([0, -0.0, 65504, 1e-8, 6.1e-5, 1 + 2**-11, 1 + 3 * 2**-11], dtype=np.float32)])
    eng = _native.Engine(0)
    y = eng.to_f16(torch.from_numpy(x).cuda()).cpu().numpy()
    np.testing.assert_array_equal(y.view(np.uint16), x.astype(np.float16).view(np.uint16))
    eng.close()


# --------------------------------------------------------------------------------------
# the C ABI's device-resident calls: per-shard search + merge == whole-corpus search
# --------------------------------------------------------------------------------------
def test_shard_search_plus_merge_equals_whole():
    import torch

    v, _ = make_corpus(9001, 1536, 1234)
    qs = make_queries(5, 1536, 1235)
    k = 32
    whole = _native.Engine(0)
    whole.upload_rows(v, 0, _native.TAVB_F32)
    dq = torch.from_numpy(qs).cuda()
    kw = whole.search_device(dq, k, 0.0)
    whole.synchronize()
    want = kw.cpu().numpy()
    parts = []
    bounds = [0, 1, 2500, 2501, 9001]
    engines = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        e = _native.Engine(0)
        e.ordinal_base = lo
        e.upload_rows(v[lo:hi], 0, _native.TAVB_F32)
        keys = e.search_device(dq, k, 0.0)
        e.synchronize()
        parts.append(keys)
        engines.append(e)
    gathered = torch.stack(parts).contiguous()
    merged = whole.merge_device(gathered)
    whole.synchronize()
    np.testing.assert_array_equal(merged.cpu().numpy(), want)
    ords, scs, cnts = _native.decode_keys(want)
    for qi in range(5):
        vo.check_topk_parity(vo.scores_full(v, qs[qi]), ords[qi, : cnts[qi]], scs[qi, : cnts[qi]], k, 0.0, referee=vo.f64_referee(v, qs[qi]))
    for e in engines + [whole]:
        e.close()


def test_sharded_searcher_on_one_rank_rccl():
    """The product's N > 1 code path on a one-rank world: DeviceShardBackend + libtavb's OWN RCCL communicator
    (tavb_comm_unique_id / tavb_comm_init / tavb_search_allgather: scan -> ncclAllGather on the context's stream -> merge kernel ->
    pinned host memory), no torch.distributed anywhere; then the same through the torch.distributed fallback the CPU test backend
    uses.  Same answer as the plain engine."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from typeagent_py_amd.sharded import DeviceShardBackend, ShardedSearcher

    v, _ = make_corpus(10_000, 1536, 8100)
    qs = make_queries(6, 1536, 8101)
    backend = DeviceShardBackend(0)
    with torch.cuda.stream(backend.stream):
        shard = torch.from_numpy(v).cuda()
    backend.set_shard(shard, row_offset=5_000_000)
    dq = torch.from_numpy(qs).cuda()
    many = make_queries(1024, 1536, 8102)
    dmany = torch.from_numpy(many).cuda()
    torch.cuda.synchronize()  # (the inputs were copied on torch's default stream, the lookups run on the backend's)

    def check(res):
        for qi in range(6):
            m = int(res.counts[qi])
            assert m == 32
            vo.check_topk_parity(vo.scores_full(v, qs[qi]), (res.ordinals[qi, :m] - 5_000_000).tolist(), res.scores[qi, :m].tolist(), 32, 0.0, referee=vo.f64_referee(v, qs[qi]))

    # 1. the library's communicator, world of one, the collective forced
    assert not dist.is_initialized()
    backend.init_comm(0, 1)
    eng = backend.engine
    eng.set_option("comm_force", 1)
    assert eng.get_option("comm_world") == 1 and eng.get_option("comm_rank") == 0
    eng.profile_enable(True)
    eng.profile_reset()
    searcher = ShardedSearcher(backend)
    check(searcher.search(dq, 32, 0.0))
    assert eng.profile_read(_native.KERNEL_EXCHANGE)[1] == 1  # the all-gather ran, inside libtavb
    # 1024 queries: the wide tile in front of the exchange
    res = searcher.search(dmany, 32, 0.0)
    res = type(res)(res.ordinals.copy(), res.scores.copy(), res.counts.copy())
    keys = eng.search_device(dmany, 32, 0.0)
    eng.synchronize()
    plain = _native.decode_keys(keys.cpu().numpy())
    np.testing.assert_array_equal(res.ordinals, plain[0])
    np.testing.assert_array_equal(res.scores, plain[1])
    # the other forms of the VectorBase-shaped front end ride the same communicator: tavb_search_subset_device + tavb_remap_key_positions +
    # tavb_allgather_merge (subset), one emit-all pass + host predicate + tavb_allgather_merge (predicate)
    from typeagent_py_amd.sharded import ShardedVectorBase

    backend.set_shard(shard, row_offset=0)
    svb = ShardedVectorBase(backend, 0, 10_000, 10_000)
    n_exchanges = eng.profile_read(_native.KERNEL_EXCHANGE)[1]
    subset = np.random.default_rng(8103).integers(-50, 10_000, size=700).tolist() + [17, 17]
    got = svb.fuzzy_lookup_embedding_in_subset(qs[1], subset, max_hits=20, min_score=0.0)
    sub_a = np.asarray(subset, dtype=np.int64)
    vo.check_topk_parity(vo.scores_full(v, qs[1])[sub_a], [r.item for r in got], [r.score for r in got], 20, 0.0, candidate_ordinals=sub_a)
    assert len(got) == 20
    pred = lambda i: i % 7 == 3
    got = svb.fuzzy_lookup_embedding(qs[2], max_hits=12, min_score=0.5, predicate=pred)
    want = vo.lookup(v, qs[2], 12, 0.5, predicate=pred)
    assert [r.item for r in got] == [i for i, _ in want]
    np.testing.assert_allclose([r.score for r in got], [s_ for _, s_ in want], atol=1e-6, rtol=0)
    msgs = svb.lookup_messages_by_embedding(qs[3], [i // 3 for i in range(10_000)], max_matches=15, threshold_score=0.0)
    assert len(msgs) <= 15 and len({m.item for m in msgs}) == len(msgs)
    assert eng.profile_read(_native.KERNEL_EXCHANGE)[1] == n_exchanges + 3
    with pytest.raises(IndexError):
        svb.fuzzy_lookup_embedding_in_subset(qs[1], [10_000], max_hits=5)
    backend.set_shard(shard, row_offset=5_000_000)
    # fault injection: the local search of this rank "fails" -- it still joins the all-gather (the exchange runs: peers are never left waiting),
    # returns ITS error, and what the merge left in the output decodes to TAVB_E_PEER: no rank can take a result that misses a shard for an answer
    n_exchanges = eng.profile_read(_native.KERNEL_EXCHANGE)[1]
    pinned = torch.empty((6, 32), dtype=torch.int64).pin_memory()
    eng.set_option("comm_fail_rank", 0)
    with pytest.raises(_native.TavbError, match="injected failure"):
        eng.search_allgather(dq, 32, 0.0, out_keys=pinned)
    eng.synchronize()
    assert eng.profile_read(_native.KERNEL_EXCHANGE)[1] == n_exchanges + 1
    assert (pinned.numpy().view(np.uint64) == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    with pytest.raises(_native.TavbError, match="rank of the collective lookup failed"):
        _native.decode_keys(pinned.numpy())
    with pytest.raises(_native.TavbError):
        searcher.search(dq, 32, 0.0)
    eng.set_option("comm_fail_rank", 3)  # another rank's number: nothing happens here
    check(searcher.search(dq, 32, 0.0))
    eng.set_option("comm_fail_rank", -1)
    check(searcher.search(dq, 32, 0.0))
    with pytest.raises(ValueError):
        eng.comm_init(b"x" * 128, 0, 1)  # one communicator per context
    eng.comm_destroy()
    assert eng.get_option("comm_world") == 0
    eng.profile_enable(False)
    backend.native_comm = False

    # 2. the torch.distributed route (what a backend without a native communicator takes)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        check(ShardedSearcher(backend, always_collective=True).search(dq, 32, 0.0))
    finally:
        dist.destroy_process_group()


def test_exchange_allocates_nothing_chunks_big_lists_and_times_out():
    """`tavb_comm_init` reserves the exchange buffers (option comm_reserve_keys): an exchange of up to that many keys allocates nothing between
    entering `tavb_search_allgather` and ncclAllGather; a bigger one travels through the same buffers in chunks of whole queries -- same
    answer; when its per-call list allocation fails (injected: comm_fail_alloc) the rank still joins EVERY chunk with TAVB_KEY_PEER_FAILED
    lists.  A rank without a usable shard joins too.  `comm_timeout_ms`: an exchange that does not complete in time (injected: the stream
    held up for 400 ms, as by a late peer) aborts the communicator and `synchronize()` raises TavbTimeout instead of waiting for ever."""
    import torch

    from typeagent_py_amd.sharded import DeviceShardBackend

    v, _ = make_corpus(20_000, 256, 8150)
    qs = make_queries(100, 256, 8151)
    backend = DeviceShardBackend(0)
    eng = backend.engine
    with torch.cuda.stream(backend.stream):
        shard = torch.from_numpy(v).cuda()
        dq = torch.from_numpy(qs).cuda()
    backend.stream.synchronize()
    backend.set_shard(shard, row_offset=1000)
    with pytest.raises(ValueError):
        eng.set_option("comm_reserve_keys", 16)  # below one list of TAVB_MAX_FUSED_K keys
    eng.set_option("comm_reserve_keys", 1024)  # 32 queries of k = 32 per chunk: 100 queries = 4 chunks
    backend.init_comm(0, 1)
    eng.set_option("comm_force", 1)
    with pytest.raises(ValueError):
        eng.set_option("comm_reserve_keys", 4096)  # read by tavb_comm_init: too late
    plain = eng.search_device(dq, 32, 0.0)
    eng.synchronize()
    plain = plain.cpu().numpy().copy()
    pinned = torch.empty((100, 32), dtype=torch.int64).pin_memory()
    eng.profile_enable(True)
    eng.profile_reset()
    eng.search_allgather(dq, 32, 0.0, out_keys=pinned)
    eng.synchronize()
    assert eng.profile_read(_native.KERNEL_EXCHANGE)[1] == 4  # four chunks
    np.testing.assert_array_equal(pinned.numpy(), plain)
    o, s_, c_ = _native.decode_keys(pinned.numpy())
    vo.check_topk_parity(vo.scores_full(v, qs[99]), (o[99, : c_[99]] - 1000).tolist(), s_[99, : c_[99]].tolist(), 32, 0.0, referee=vo.f64_referee(v, qs[99]))
    # a list that fits the reserved buffer: one all-gather, nothing allocated on the way (the injected allocation failure is never reached)
    plain32 = eng.search_device(dq[:32], 32, 0.0)  # (a 32-query batch rides another tile than a 100-query one: its own float32 sums)
    eng.synchronize()
    eng.set_option("comm_fail_alloc", 1)
    eng.profile_reset()
    small = torch.empty((32, 32), dtype=torch.int64).pin_memory()
    eng.search_allgather(dq[:32], 32, 0.0, out_keys=small)
    eng.synchronize()
    assert eng.profile_read(_native.KERNEL_EXCHANGE)[1] == 1
    np.testing.assert_array_equal(small.numpy(), plain32.cpu().numpy())
    # the big one with its list allocation failing: the rank's own error, AND all four chunks of the exchange ran with the failure key
    eng.profile_reset()
    pinned.zero_()
    with pytest.raises(_native.TavbError, match="injected failure of the list allocation"):
        eng.search_allgather(dq, 32, 0.0, out_keys=pinned)
    eng.synchronize()
    assert eng.profile_read(_native.KERNEL_EXCHANGE)[1] == 4
    assert (pinned.numpy().view(np.uint64) == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    with pytest.raises(_native.TavbError, match="rank of the collective lookup failed"):
        _native.decode_keys(pinned.numpy())
    eng.set_option("comm_fail_alloc", 0)
    eng.search_allgather(dq, 32, 0.0, out_keys=pinned)
    eng.synchronize()
    np.testing.assert_array_equal(pinned.numpy(), plain)  # and the next lookup lines up
    # a shard whose ordinals do not fit the keys: a local failure like any other -- the rank joins the exchange, then reports it
    backend.set_shard(shard, row_offset=0xFFFFFFFF - 100)
    eng.profile_reset()
    with pytest.raises(_native.TavbError, match="32-bit ordinals"):
        eng.search_allgather(dq, 32, 0.0, out_keys=pinned)
    eng.synchronize()
    assert eng.profile_read(_native.KERNEL_EXCHANGE)[1] == 4
    assert (pinned.numpy().view(np.uint64) == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    backend.set_shard(shard, row_offset=1000)
    eng.profile_enable(False)
    # the timeout: a generous one changes nothing ...
    eng.set_option("comm_timeout_ms", 5000)
    eng.search_allgather(dq, 32, 0.0, out_keys=pinned)
    eng.synchronize()
    np.testing.assert_array_equal(pinned.numpy(), plain)
    # ... an exchange held up for 400 ms against a 50 ms limit: TavbTimeout, the communicator is gone, the stream drains, the context works on
    import time

    eng.set_option("comm_timeout_ms", 50)
    eng.set_option("comm_stall_ms", 400)
    eng.search_allgather(dq, 32, 0.0, out_keys=pinned)
    t0 = time.perf_counter()
    with pytest.raises(_native.TavbTimeout, match="did not complete within"):
        eng.synchronize()
    assert time.perf_counter() - t0 < 5.0
    assert eng.get_option("comm_world") == 0 and eng.get_option("comm_stall_ms") == 0
    eng.synchronize()  # nothing in flight any more
    again = eng.search_device(dq, 32, 0.0)
    eng.synchronize()
    np.testing.assert_array_equal(again.cpu().numpy(), plain)
    # rejoin
    backend.init_comm(0, 1)
    eng.set_option("comm_force", 1)
    eng.set_option("comm_timeout_ms", 0)
    eng.search_allgather(dq, 32, 0.0, out_keys=pinned)
    eng.synchronize()
    np.testing.assert_array_equal(pinned.numpy(), plain)
    eng.comm_destroy()


def test_sharded_vectorbase_storage_methods_on_the_device():
    """ShardedVectorBase.add_embedding(s) / serialize / deserialize / clear over a DeviceShardBackend (one rank: the same code every rank of
    a job runs; the world-2 form of it runs under gloo in tests/test_sharded_gloo.py): appends reach the device shard incrementally -- a
    shard adopted from the caller's tensor is copied first, the caller's tensor is never written -- and lookups see them under their
    global ordinals."""
    import torch

    from typeagent_py_amd.sharded import DeviceShardBackend, ShardedVectorBase

    v, q = make_corpus(3000, 256, 8400)
    extra, _ = make_corpus(40, 256, 8401)
    extra[11] = q
    backend = DeviceShardBackend(0)
    with torch.cuda.stream(backend.stream):
        shard = torch.from_numpy(v).cuda()
    backend.stream.synchronize()
    backend.set_shard(shard, row_offset=0)
    svb = ShardedVectorBase(backend, 0, 3000, 3000)
    first = svb.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(v, q), *items_scores(first), 10, 0.0, referee=vo.f64_referee(v, q))
    svb.add_embeddings(None, extra[:30])
    svb.add_embedding("x", extra[30])
    svb.add_embeddings(["k"] * 9, extra[31:])
    assert len(svb) == 3040 and svb.local_rows == 3040
    grown = np.concatenate([v, extra])
    res = svb.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.0)
    assert res[0].item == 3011 and abs(res[0].score - 1.0) < 1e-6
    vo.check_topk_parity(vo.scores_full(grown, q), *items_scores(res), 10, 0.0, referee=vo.f64_referee(grown, q))
    np.testing.assert_array_equal(shard.cpu().numpy(), v)  # the adopted tensor was not touched
    np.testing.assert_array_equal(svb.serialize(), grown)
    batch = svb.fuzzy_lookup_embeddings(np.stack([q, grown[5]]), max_hits=5)
    assert batch[0][0].item == 3011 and batch[1][0].item == 5
    svb.deserialize(grown[1000:2000], dtype="fp16")
    assert len(svb) == 1000 and svb.row_offset == 0
    seen = _f16(grown[1000:2000])
    res = svb.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(seen, q), *items_scores(res), 10, 0.0, referee=vo.f64_referee(seen, q))
    np.testing.assert_array_equal(svb.serialize(), seen)
    svb.add_embeddings(None, extra[11:12])
    assert svb.fuzzy_lookup_embedding(q, max_hits=1)[0].item == 1000
    svb.clear()
    assert len(svb) == 0 and svb.fuzzy_lookup_embedding(q, max_hits=3) == []
    svb.add_embeddings(None, extra)  # an index grown from nothing
    assert svb.fuzzy_lookup_embedding(q, max_hits=1)[0].item == 11


def test_fused_multi_index_query_equals_separate_calls():
    """cfg5: T term lookups (k=50 @0.85) + message re-rank (k=25 @0.7, full scan and subset) + thread lookup
    (k=10 @0.7) in one submission == the same lookups issued one by one == the oracle."""
    import torch

    from typeagent_py_amd.fused import FusedIndexQuery

    dim = 1536
    terms, _ = make_corpus(30_000, dim, 9100)
    msgs, _ = make_corpus(20_000, dim, 9101)
    thr, _ = make_corpus(1_000, dim, 9102)
    rng = np.random.default_rng(9103)
    # term queries close to existing rows so that min_score 0.85 keeps something
    tq = np.stack([terms[i] + 0.25 * rng.standard_normal(dim).astype(np.float32) / np.sqrt(dim) for i in (5, 777, 12_345, 29_999)])
    tq /= np.linalg.norm(tq, axis=1, keepdims=True)
    mq = msgs[4242] + 0.6 * rng.standard_normal(dim).astype(np.float32) / np.sqrt(dim)
    mq /= np.linalg.norm(mq)
    hq = thr[17] + 0.6 * rng.standard_normal(dim).astype(np.float32) / np.sqrt(dim)
    hq /= np.linalg.norm(hq)
    subset = subset_choice(20_000, 1000, 99) + [4242]

    fq = FusedIndexQuery(0)
    fq.set_corpus("terms", torch.from_numpy(terms).cuda())
    fq.set_corpus("messages", torch.from_numpy(msgs).cuda())
    fq.set_corpus("threads", torch.from_numpy(thr).cuda())
    torch.cuda.synchronize()
    full = fq.run(tq, mq, hq)
    sub = fq.run(tq, mq, hq, message_subset=subset)

    vt, vm, vh = new_vb(terms), new_vb(msgs), new_vb(thr)
    for i in range(4):
        single = vt.fuzzy_lookup_embedding(tq[i], max_hits=50, min_score=0.85)
        assert [r.item for r in full.terms[i]] == [r.item for r in single] and len(single) >= 1
        np.testing.assert_allclose([r.score for r in full.terms[i]], [r.score for r in single], atol=2e-7, rtol=0)
        vo.check_topk_parity(vo.scores_full(terms, tq[i]), *items_scores(full.terms[i]), 50, 0.85, referee=vo.f64_referee(terms, tq[i]))
        assert [r.item for r in sub.terms[i]] == [r.item for r in single]
    single = vm.fuzzy_lookup_embedding(mq, max_hits=25, min_score=0.7)
    assert [r.item for r in full.messages] == [r.item for r in single] and single[0].item == 4242
    vo.check_topk_parity(vo.scores_full(msgs, mq), *items_scores(full.messages), 25, 0.7, referee=vo.f64_referee(msgs, mq))
    single = vm.fuzzy_lookup_embedding_in_subset(mq, subset, max_hits=25, min_score=0.7)
    assert [r.item for r in sub.messages] == [r.item for r in single] and 4242 in [r.item for r in sub.messages]
    single = vh.fuzzy_lookup_embedding(hq, max_hits=10, min_score=0.7)
    assert [r.item for r in full.threads] == [r.item for r in single] and single[0].item == 17
    # missing pieces are simply skipped
    only_terms = fq.run(tq[:2])
    assert len(only_terms.terms) == 2 and only_terms.messages == [] and only_terms.threads == []
    # ... also when the call before (same shape: the result buffer is reused, its keys written by the kernels straight into pinned memory) had them
    fq.run(tq, mq, hq)
    terms_again = fq.run(tq)
    assert terms_again.messages == [] and terms_again.threads == []
    assert [[r.item for r in t] for t in terms_again.terms] == [[r.item for r in t] for t in full.terms]
    no_terms = fq.run(np.zeros((0, dim), np.float32), mq, hq)
    assert no_terms.terms == [] and [r.item for r in no_terms.messages] == [r.item for r in full.messages]
    assert [r.item for r in no_terms.threads] == [r.item for r in full.threads]


@pytest.mark.slow
def test_cfg3_full_size_batch_against_chunked_oracle():
    """BASELINE config 3 at full size (10M x 1536 fp16, 1024 arbitrary fp32 queries, top-32, MFMA filter + rescoring):
    16 sampled queries are checked against the oracle over the WHOLE corpus (delivered in 1M-row chunks, widened to
    fp32), 8 of them also against the streaming kernel, plus size-independent properties on all 1024 answers and the
    near-tie count of the reference ranking at this size."""
    from bench import ORACLE_CHUNK, host_queries, make_device_corpus

    rows, dim, nq, k = 10_000_000, 1536, 1024, 32
    eng = _native.Engine(0)
    corpus = make_device_corpus(eng, rows, dim, 10_043, "fp16")
    eng.set_corpus_tensor(corpus)
    eng.profile_enable(True)
    eng.profile_reset()
    qs = host_queries(nq, dim, 4242)
    planted = {0: 123_456, 7: 9_999_999, 1000: 5_000_000}  # query index -> row whose (fp16) values become the query
    for qi, r in planted.items():
        qs[qi] = corpus[r].float().cpu().numpy()
    ords, scs, cnts = eng.search_batch(qs, k, np.float32(0.0))
    assert eng.profile_read(_native.KERNEL_MFMA)[1] >= 1 and eng.get_option("last_flagged") == 0
    assert np.all(cnts == k)
    assert np.all(np.diff(scs, axis=1) <= 0)  # sorted best first
    assert np.all((ords >= 0) & (ords < rows))
    assert all(len(set(ords[i].tolist())) == k for i in range(nq))  # no duplicates
    for qi, r in planted.items():
        assert ords[qi, 0] == r and abs(scs[qi, 0] - 1.0) < 2e-3
    sample = sorted(set(np.linspace(0, nq - 1, 16).astype(int).tolist()) | {0, 7, 1000})
    # the same queries one at a time through the streaming kernel (independent code path, fp32 query x fp16 row)
    for qi in sample[:8]:
        o1, s1 = eng.search(qs[qi], k, np.float32(0.0))
        assert eng.get_option("last_tier") in (1, 2, 3)
        np.testing.assert_array_equal(o1, ords[qi])
        np.testing.assert_allclose(s1, scs[qi], atol=3e-7, rtol=0)
    ref, referee = vo.scores_full_chunked_refereed((corpus[lo : lo + ORACLE_CHUNK].float().cpu().numpy() for lo in range(0, rows, ORACLE_CHUNK)), qs[sample],
                                                   [ords[qi] for qi in sample], keep=k + 256)
    near_total = permuted = inv_gpu = inv_ref = 0
    for j, qi in enumerate(sample):
        rep, near = vo.check_topk_parity_large(ref[j], ords[qi].tolist(), scs[qi].tolist(), k, 0.0, referee=referee.for_query(j))
        near_total += near
        permuted += rep.tie_permuted_positions
        inv_gpu += rep.gpu_inversions_vs_f64
        inv_ref += rep.reference_inversions_vs_f64
    print(f"cfg3 full size: {len(sample)} queries, near-tie pairs in the reference top-{k}: {near_total}, positions permuted inside them: {permuted}; "
          f"pairs ordered against the float64 truth: device {inv_gpu}, reference {inv_ref}")
    assert permuted <= near_total
    eng.close()


def _two_rank_worker(rank, world, port, ret):
    import os
    import sys

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.synth import make_corpus, make_queries
        from typeagent_py_amd.sharded import DeviceShardBackend, ShardedSearcher, shard_range

        v, _ = make_corpus(30_001, 1536, 8200)
        qs = make_queries(7, 1536, 8201)
        lo, hi = shard_range(len(v), world, rank)
        backend = DeviceShardBackend(0)  # both ranks share GPU 0 here; RCCL refuses that, so the exchange goes over gloo
        with torch.cuda.stream(backend.stream):
            shard = torch.from_numpy(v[lo:hi]).cuda()
        backend.set_shard(shard, row_offset=lo)

        def gather_over_gloo(local):
            host = local.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host)
            return torch.stack(parts).contiguous().cuda()

        searcher = ShardedSearcher(backend, gather_fn=gather_over_gloo)
        res = searcher.search(torch.from_numpy(qs).cuda(), 32, 0.0)
        ret[rank] = (res.ordinals.copy(), res.scores.copy(), res.counts.copy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_device_kernels_gloo_exchange():
    """Two processes, each holding half of the rows on the device with its ordinal offset baked into the
    keys; per-shard HIP search + merge kernel on both ranks; only the exchange itself is gloo instead of
    RCCL (two ranks cannot share one GPU under RCCL).  Both ranks must return the whole-corpus answer."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_two_rank_worker, args=(2, port, ret), nprocs=2, join=True)
    v, _ = make_corpus(30_001, 1536, 8200)
    qs = make_queries(7, 1536, 8201)
    for qi in range(7):
        np.testing.assert_array_equal(ret[0][0][qi], ret[1][0][qi])
        np.testing.assert_array_equal(ret[0][1][qi], ret[1][1][qi])
        rep = vo.check_topk_parity(vo.scores_full(v, qs[qi]), ret[0][0][qi].tolist(), ret[0][1][qi].tolist(), 32, 0.0, referee=vo.f64_referee(v, qs[qi]))
        assert rep.ordinals_bit_exact
    assert ret[0][0].max() > 15_001  # hits from the second shard carry their global ordinals


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_device_resident_load_paths_stream_without_a_host_matrix(tmp_path, dtype):
    """`_embeddings.bin` and SQLite BLOB reloads stream chunk by chunk through the pinned staging ring into the
    capacity-doubling DEVICE corpus (on-device fp16 conversion); no host matrix is built.  Lookups match the oracle,
    serialize() copies back on demand (and the index is host-authoritative from then on)."""
    import sqlite3

    from typeagent_py_amd.adapters import load_embeddings_bin, load_sqlite_embeddings

    rng = np.random.default_rng(77)
    related, _ = make_corpus(5_003, 384, 7701)
    messages, q = make_corpus(20_011, 384, 7702)
    path = tmp_path / "x_embeddings.bin"
    with open(path, "wb") as f:  # knowpro/serialization.py:84-98: related-term rows, then message rows
        f.write(related.astype("<f4").tobytes())
        f.write(messages.astype("<f4").tobytes())
    rv = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=dtype, keep_host_copy=False)
    mv = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=dtype, keep_host_copy=False)
    assert load_embeddings_bin(str(path), 384, len(related), len(messages), rv, mv, chunk_rows=3000) == (len(related), len(messages))
    assert len(mv) == len(messages) and mv._host.shape[0] == 0 and mv._device_only is not None  # nothing on the host
    want = messages.astype(np.float16).astype(np.float32) if dtype == "fp16" else messages
    res = mv.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(want, q), *items_scores(res), 32, 0.0, referee=vo.f64_referee(want, q))
    assert mv._host.shape[0] == 0  # a lookup does not materialise it either
    np.testing.assert_array_equal(mv.serialize(), want)  # ... serialize() does (fp16 storage: the widened values)
    mv.add_embeddings(None, related[:5])  # host-authoritative now: appends behave as always
    assert len(mv) == len(messages) + 5 and mv.fuzzy_lookup_embedding(related[2], max_hits=1)[0].item == len(messages) + 2
    # SQLite BLOB column, fetched 1000 rows at a time (storage/sqlite/messageindex.py:33-45)
    db = sqlite3.connect(str(tmp_path / "c.db"))
    db.execute("CREATE TABLE MessageTextIndex (msg_id INTEGER, chunk_ordinal INTEGER, embedding BLOB NOT NULL, index_position INTEGER)")
    db.executemany("INSERT INTO MessageTextIndex VALUES (?, ?, ?, ?)", [(i // 2, i % 2, row.tobytes(), i) for i, row in enumerate(related)])
    db.commit()
    sv = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=dtype, keep_host_copy=False)
    load_sqlite_embeddings(db, sv, fetch_rows=1000)
    assert len(sv) == len(related) and sv._host.shape[0] == 0
    want_r = related.astype(np.float16).astype(np.float32) if dtype == "fp16" else related
    vo.check_topk_parity(vo.scores_full(want_r, q), *items_scores(sv.fuzzy_lookup_embedding(q, max_hits=10, min_score=0.0)), 10, 0.0, referee=vo.f64_referee(want_r, q))
    sv.add_embedding(None, related[0])  # single rows stream too
    assert len(sv) == len(related) + 1 and sv._host.shape[0] == 0


def test_appending_to_an_adopted_tensor_copies_it_first():
    import torch

    v, q = make_corpus(4_000, 384, 7800)
    t = torch.from_numpy(np.concatenate([v, np.zeros((100, 384), np.float32)])).cuda()  # spare capacity behind the 4000 rows
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), keep_host_copy=False)
    vb.adopt_device_corpus(t, rows=4_000)
    vb.add_embeddings(None, v[:10] * 2.0)
    assert len(vb) == 4_010 and float(t[4_000:].abs().sum()) == 0.0  # the caller's tensor was not written to
    assert [r.item for r in vb.fuzzy_lookup_embedding(v[3], max_hits=2, min_score=0.0)] == [3, 4_003]  # the 2x row clips to 1.0 too: a tie, ascending ordinal
    vb2 = new_vb()
    vb2.adopt_device_corpus(t, rows=4_000)
    vb2.add_embedding(None, v[0])  # host-copy mode: materialises, then appends on the host
    assert len(vb2) == 4_001 and float(t[4_000:].abs().sum()) == 0.0


def test_device_only_corpus_and_lazy_host_copy():
    import torch

    v, q = make_corpus(3000, 1536, 4321)
    vb = new_vb()
    vb.adopt_device_corpus(torch.from_numpy(v).cuda())
    assert len(vb) == 3000 and vb._embedding_size == 1536
    res = vb.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    vo.check_topk_parity(vo.scores_full(v, q), *items_scores(res), 32, 0.0, referee=vo.f64_referee(v, q))
    np.testing.assert_array_equal(vb.serialize(), v)  # copied back on demand


def test_profile_counters_count_launches():
    v, q = make_corpus(20000, 1536, 55)
    vb = new_vb(v)
    eng = vb.engine
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(3):
        vb.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    ms, n = eng.profile_read(_native.KERNEL_SCAN)
    assert n == 3 and ms > 0
    assert eng.profile_read(_native.KERNEL_MERGE)[1] == 0  # a corpus this small (123 MB) takes the one-launch path: lists merged on the host
    eng.set_option("small_direct_bytes", 0)
    eng.profile_reset()
    for _ in range(3):
        vb.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    assert eng.profile_read(_native.KERNEL_SCAN)[1] == 3
    ms2, n2 = eng.profile_read(_native.KERNEL_MERGE)
    assert n2 == 3 and ms2 > 0
    eng.profile_enable(False)


def test_device_memory_does_not_grow_over_many_lookups_of_the_same_shapes():
    """Workspaces of the library grow to the largest shape asked for and stay there: a few hundred lookups of a fixed set of shapes (single
    query, 40 / 70 / 300-query batches, subsets, message re-rank, appends in between) leave the free device memory where the first round put it."""
    import torch

    v, _ = make_corpus(40_000, 1536, 8800)
    qs = make_queries(300, 1536, 8801)
    vb = new_vb(v[:30_000], dtype="fp16")
    vb.set_row_messages(np.arange(40_000) // 3)
    sub = list(range(0, 30_000, 7))

    def one_round(i):
        vb.fuzzy_lookup_embedding(qs[i % 300], max_hits=10, min_score=0.0)
        vb.fuzzy_lookup_embeddings(qs[:40], max_hits=32, min_score=0.0)
        vb.fuzzy_lookup_embeddings(qs[:70], max_hits=50, min_score=0.5)
        vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0, as_arrays=True)
        vb.fuzzy_lookup_embedding_in_subset(qs[(i + 1) % 300], sub, max_hits=25, min_score=0.0)
        vb.lookup_messages_by_embedding(qs[(i + 2) % 300], 25, 0.0)

    for i in range(3):
        one_round(i)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(60):
        one_round(i)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), (free0, free1)
    # appends: the first one moves the corpus to a buffer of twice the size (+ staging ring, conversion scratch: one-off); the following ones
    # fill it -- no growth per append
    marks = []
    for j in range(20):
        vb.add_embeddings(None, v[30_000 + 500 * j : 30_000 + 500 * (j + 1)])
        one_round(j)
        if j in (4, 19):
            torch.cuda.synchronize()
            marks.append(torch.cuda.mem_get_info()[0])
    assert free1 - marks[0] < (2 << 30), (free1, marks)  # one-off, bounded
    assert marks[0] - marks[1] < (64 << 20), marks       # 15 more appends: nothing more
    res = vb.fuzzy_lookup_embedding(v[39_999], max_hits=1)
    assert res[0].item == 39_999


def test_destroying_a_context_returns_its_workspaces_and_the_shadow():
    """Every workspace of a context -- the fp16 shadow of an fp32 corpus (half the corpus' bytes) and the padded query copy included -- goes
    back to the device when the context is destroyed: engines created, used with wide batches and closed in a loop do not eat the device."""
    import torch

    v, _ = make_corpus(60_000, 1536, 8810)  # 369 MB of fp32 rows: a 184 MB shadow per context
    v8, _ = make_corpus(20_000, 1000, 8812)
    qs = make_queries(70, 1536, 8811)
    qs8 = make_queries(70, 1000, 8813)

    def use_and_close():
        vb = new_vb(v)
        vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
        assert vb.engine.get_option("last_tier") == 4 and vb.engine.get_option("last_shadow") == 1
        vb.engine.close()
        odd = new_vb(v8, dtype="fp16")
        odd.fuzzy_lookup_embeddings(qs8, max_hits=32, min_score=0.0)
        assert odd.engine.get_option("last_tier") == 4
        odd.engine.close()
        del vb, odd
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    use_and_close()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(4):
        use_and_close()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (96 << 20), (free0, free1)  # four more shadows would be 740 MB
