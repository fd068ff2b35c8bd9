"""CPU suite: the numpy oracle (oracle/vectorbase_oracle.py) against the committed
golden vectors, which were produced by the VERBATIM reference class
(tests/golden/make_golden.py).  This is what pins the oracle on machines where
/root/reference does not exist."""

import hashlib

import numpy as np
import pytest

from oracle import vectorbase_oracle as vo
from tests.synth import explicit_case_arrays, make_corpus, subset_choice

MAX_CPU_ROWS = 100_000  # the 1M-row case is exercised by the GPU suite (it needs 12 GB of host RAM)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _same(got, expect, exact_items=True):
    items = [i for i, _ in got]
    scores = [s for _, s in got]
    if exact_items:
        assert items == expect["items"]
        assert scores == pytest.approx(expect["scores"], abs=0, rel=0)
    else:
        assert len(items) == len(expect["items"])
        np.testing.assert_allclose(scores, expect["scores"], atol=1e-6, rtol=0)


def test_explicit_cases(golden):
    for case in golden["explicit"]:
        v, q = explicit_case_arrays(case)
        kw = dict(case["args"])
        with np.errstate(invalid="ignore"):
            if "subset" in case:
                if case.get("raises") == "IndexError":
                    with pytest.raises(IndexError):
                        vo.lookup_in_subset(v, q, case["subset"], **kw)
                    continue
                got = vo.lookup_in_subset(v, q, case["subset"], **kw)
            elif "predicate_mod" in case:
                m, r = case["predicate_mod"]
                got = vo.lookup(v, q, predicate=lambda i: i % m == r, **kw)
            else:
                got = vo.lookup(v, q, **kw)
        _same(got, case["expect"])


def test_reference_known_answers_are_in_the_goldens(golden):
    by_name = {c["name"]: c for c in golden["explicit"]}
    ka = by_name["ref_test_normalized_score_scale"]  # reference tests/test_vectorbase.py:239-252
    assert ka["expect"]["items"] == [0, 1, 2] and ka["expect"]["scores"] == [1.0, 0.5, 0.0]
    two = by_name["ref_test_benchmark_embeddings_two_rows"]  # tests/test_benchmark_embeddings.py:229-277
    assert two["expect"]["items"][0] == 1 and len(two["expect"]["items"]) == 2 and two["expect"]["scores"][1] == 0.5


@pytest.mark.parametrize("idx", range(8))
def test_seeded_cases(golden, idx):
    entry = golden["seeded"][idx]
    if entry["n"] > MAX_CPU_ROWS:
        pytest.skip("covered by the GPU suite")
    v, q = make_corpus(entry["n"], entry["d"], entry["seed"])
    assert _sha(v) == entry["corpus_sha256"], "numpy RNG stream drifted: regenerate goldens"
    assert _sha(q) == entry["query_sha256"]
    for run in entry["runs"]:
        kw = dict(run["args"])
        if run["kind"] == "full":
            got = vo.lookup(v, q, **kw)
        elif run["kind"] == "subset":
            sub = subset_choice(entry["n"], run["subset_args"]["size"], run["subset_args"]["seed"])
            assert _sha(np.asarray(sub, dtype=np.int64)) == run["subset_sha256"]
            got = vo.lookup_in_subset(v, q, sub, **kw)
        elif run["kind"] == "full_f16_corpus_f32_query":
            got = vo.lookup(v.astype(np.float16).astype(np.float32), q, **kw)
        elif run["kind"] == "full_f16_corpus_f16_query":
            got = vo.lookup(v.astype(np.float16).astype(np.float32), q.astype(np.float16).astype(np.float32), **kw)
        else:
            raise AssertionError(run["kind"])
        # same numpy, same BLAS => bit-identical to the verbatim reference
        _same(got, run["expect"])
        # and the parity checker accepts the reference's own answer
        if run["kind"] == "full":
            sc = vo.scores_full(v, q)
            k = 10 if kw["max_hits"] is None else kw["max_hits"]
            ms = 0.0 if kw["min_score"] is None else kw["min_score"]
            rep = vo.check_topk_parity(sc, run["expect"]["items"], run["expect"]["scores"], k, ms)
            assert rep.k_returned == len(run["expect"]["items"])


def test_settings_defaults_golden(golden):
    # reference tests/test_vectorbase.py:280-325
    d = golden["settings_defaults"]
    assert d["text-embedding-3-large"]["min_score"] == 0.74
    assert d["text-embedding-3-small"]["min_score"] == 0.73
    assert d["text-embedding-ada-002"]["min_score"] == 0.93
    assert d["custom-embedding-model"]["min_score"] == 0.85
    assert all(v["max_matches"] is None and v["batch_size"] == 8 for v in d.values())


def test_parity_checker_rejects_wrong_answers():
    v, q = make_corpus(2000, 64, 5)
    sc = vo.scores_full(v, q)
    good = vo.lookup(v, q, 10, 0.0)
    items = [i for i, _ in good]
    scores = [s for _, s in good]
    vo.check_topk_parity(sc, items, scores, 10, 0.0)
    with pytest.raises(AssertionError):  # swapped ranks
        vo.check_topk_parity(sc, [items[1], items[0]] + items[2:], [scores[1], scores[0]] + scores[2:], 10, 0.0)
    with pytest.raises(AssertionError):  # a worse row smuggled in
        worst = int(np.argmin(sc))
        vo.check_topk_parity(sc, items[:-1] + [worst], scores[:-1] + [float(sc[worst])], 10, 0.0)
    with pytest.raises(AssertionError):  # score off by more than 1e-5
        vo.check_topk_parity(sc, items, [s + 3e-5 for s in scores], 10, 0.0)
    with pytest.raises(AssertionError):  # too few
        vo.check_topk_parity(sc, items[:5], scores[:5], 10, 0.0)


def test_parity_checker_tolerates_near_ties():
    sc = np.array([0.9, 0.8, 0.8 + 2.0**-24, 0.1], dtype=np.float32)
    rep = vo.check_topk_parity(sc, [0, 1, 2], [0.9, 0.8, 0.8], 3, 0.0)
    assert rep.tie_permuted_positions == 2 and not rep.ordinals_bit_exact
    rep = vo.check_topk_parity(sc, [0, 2, 1], [0.9, float(sc[2]), 0.8], 3, 0.0)
    assert rep.ordinals_bit_exact


def _clustered_case(seed=7, n=4000, d=256, members=60):
    """A query next to a pack of near-duplicate rows: the best `members` scores lie within a few 1e-7 of one another."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, d)).astype(np.float32)
    centre = v[0] / np.linalg.norm(v[0])
    v[:members] = centre[None, :] + 1e-3 * rng.standard_normal((members, d)).astype(np.float32) / np.sqrt(d)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = centre + 0.02 * rng.standard_normal(d).astype(np.float32) / np.sqrt(d)
    q = (q / np.linalg.norm(q)).astype(np.float32)
    return v.astype(np.float32), q


def _other_fp32_answer(v, q, k):
    """What a second, equally legitimate float32 arithmetic returns: per-row dot products summed in another order (blocks of 16)."""
    prod = v * q[None, :]
    dots = prod.reshape(len(v), -1, 16).sum(axis=2, dtype=np.float32).sum(axis=1, dtype=np.float32)
    sc = vo.cosine_to_score(dots).astype(np.float32)
    order = np.lexsort((np.arange(len(v)), -sc.astype(np.float64)))[:k]
    return order.tolist(), sc[order].tolist(), sc


def test_float64_referee_accepts_another_summation_order_and_counts_inversions():
    v, q = _clustered_case()
    ref = vo.scores_full(v, q)
    items, scores, _ = _other_fp32_answer(v, q, 32)
    rep = vo.check_topk_parity(ref, items, scores, 32, 0.0, referee=vo.f64_referee(v, q))
    assert rep.refereed and rep.noise_ref > 0 and rep.noise_gpu > 0
    assert rep.tie_width == pytest.approx(2 * (rep.noise_ref + rep.noise_gpu))
    assert rep.tie_width < 16 * 2.0**-24  # narrower than the hand-set width it replaces
    assert rep.max_permuted_gap <= rep.tie_width
    assert rep.exact_positions + rep.tie_permuted_positions == 32
    # the float64 answer itself has no inversions and is accepted as well
    t = vo.scores_f64(v, q)
    best = np.lexsort((np.arange(len(v)), -t))[:32]
    rep64 = vo.check_topk_parity(ref, best.tolist(), t[best].astype(np.float32).tolist(), 32, 0.0, referee=vo.f64_referee(v, q))
    assert rep64.gpu_inversions_vs_f64 == 0 and rep64.reference_inversions_vs_f64 >= 0


def test_float64_referee_rejects_a_swap_wider_than_the_measured_noise():
    v, q = _clustered_case()
    ref = vo.scores_full(v, q)
    t = vo.scores_f64(v, q)
    good = vo.lookup(v, q, 32, 0.0)
    items, scores = [i for i, _ in good], [s for _, s in good]
    vo.check_topk_parity(ref, items, scores, 32, 0.0, referee=vo.f64_referee(v, q))
    # move a row up over a float64 gap of 5e-7 .. 8e-7: inside round 3's hand-set 16 * 2^-24 = 9.5e-7 at score ~1, outside what the two
    # measured noises (~1e-7 each) can explain
    tt = t[items]
    pair = next(((a, b) for a in range(32) for b in range(a + 1, 32) if 5e-7 < tt[a] - tt[b] < 8e-7), None)
    assert pair is not None
    a, b = pair
    bad_items = items[:a] + [items[b]] + items[a:b] + items[b + 1:]
    bad_scores = scores[:a] + [scores[a]] + scores[a:b] + scores[b + 1:]  # keeps the list descending and every score within 1e-5
    vo.check_topk_parity(ref, bad_items, bad_scores, 32, 0.0, tie_eps=16 * 2.0**-24)  # what round 3's hand-set width let through
    # a list sorted by its own scores can only swap rows over a gap <= its own error: the lie shows up as device noise (or, with honest
    # scores, as a list that is not descending)
    with pytest.raises(AssertionError, match="noisier|near-tie width"):
        vo.check_topk_parity(ref, bad_items, bad_scores, 32, 0.0, referee=vo.f64_referee(v, q))
    honest = scores[:a] + [scores[b]] + scores[a:b] + scores[b + 1:]
    with pytest.raises(AssertionError, match="descending"):
        vo.check_topk_parity(ref, bad_items, honest, 32, 0.0, referee=vo.f64_referee(v, q))


def test_float64_referee_caps_the_device_noise():
    v, q = _clustered_case()
    ref = vo.scores_full(v, q)
    good = vo.lookup(v, q, 32, 0.0)
    items = [i for i, _ in good]
    sloppy = [s - 3e-6 for _, s in good]  # inside the 1e-5 score tolerance, ordered, but 20x noisier than the reference (and than 256 float32 additions explain)
    vo.check_topk_parity(ref, items, sloppy, 32, 0.0)  # (the constant rule alone cannot see it)
    with pytest.raises(AssertionError, match="noisier"):
        vo.check_topk_parity(ref, items, sloppy, 32, 0.0, referee=vo.f64_referee(v, q))


def test_float64_referee_detects_an_omitted_row_and_handles_subsets_and_chunks():
    v, q = _clustered_case()
    ref = vo.scores_full(v, q)
    t = vo.scores_f64(v, q)
    good = vo.lookup(v, q, 10, 0.0)
    items, scores = [i for i, _ in good], [s for _, s in good]
    far = int(np.argsort(-t)[40])  # a member of the pack, but > the near-tie width below rank 10
    assert t[items[-1]] - t[far] > 1e-6
    with pytest.raises(AssertionError):
        vo.check_topk_parity(ref, items[:-1] + [far], scores[:-1] + [float(ref[far])], 10, 0.0, referee=vo.f64_referee(v, q))
    # subset form: positions index the subset
    sub = np.concatenate([np.arange(0, 60, 2), np.arange(100, 400)])
    got = vo.lookup_in_subset(v, q, sub.tolist(), 10, 0.0)
    rep = vo.check_topk_parity(ref[sub], [i for i, _ in got], [s for _, s in got], 10, 0.0, candidate_ordinals=sub, referee=vo.f64_referee(v[sub], q))
    assert rep.refereed and rep.exact_positions == 10
    # chunked form: the referee filled while the chunks pass by gives the same verdicts as the whole-matrix one
    items32, scores32, _ = _other_fp32_answer(v, q, 32)
    qs = np.stack([q, q])
    ref2, cr = vo.scores_full_chunked_refereed([v[:1500], v[1500:2600], v[2600:]], qs, [items32, items32], keep=32 + 64)
    rep_c, near = vo.check_topk_parity_large(ref2[1], items32, scores32, 32, 0.0, margin=64, referee=cr.for_query(1))
    rep_w = vo.check_topk_parity(ref2[1], items32, scores32, 32, 0.0, referee=vo.f64_referee(v, q))  # (the same float32 reference scores: sgemm's)
    assert (rep_c.tie_permuted_positions, rep_c.gpu_inversions_vs_f64) == (rep_w.tie_permuted_positions, rep_w.gpu_inversions_vs_f64)
    assert near >= rep_c.tie_permuted_positions // 2
    # chunked form of a SUBSET search (bench.py cfg3_subset): the rows that can matter are the best of the subset in each chunk, which the
    # chunk's own best rows need not contain -- `restrict` keeps the truth for them; same verdict as the whole-matrix referee on the subset
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(len(v), size=len(v) // 10, replace=False))
    got = vo.lookup_in_subset(v, q, sub.tolist(), 10, 0.0)
    pos_of = {int(o): i for i, o in enumerate(sub)}
    ref3, cr3 = vo.scores_full_chunked_refereed([v[:1500], v[1500:2600], v[2600:]], qs, [[i for i, _ in got]] * 2, keep=10 + 64, restrict=sub)
    truth = cr3.for_query(0)

    def sub_truth(p_):
        return truth(sub[np.asarray(p_)])
    sub_truth.dim = v.shape[1]
    rep_s, _ = vo.check_topk_parity_large(ref3[0][sub], [pos_of[i] for i, _ in got], [s for _, s in got], 10, 0.0, margin=64, referee=sub_truth)
    assert rep_s.refereed and rep_s.exact_positions + rep_s.tie_permuted_positions == 10
    with pytest.raises(KeyError):  # without `restrict` the subset's best rows of a chunk are not in the table
        _, cr4 = vo.scores_full_chunked_refereed([v[:1500], v[1500:2600], v[2600:]], qs, [[i for i, _ in got]] * 2, keep=10 + 64)
        t4 = cr4.for_query(0)

        def sub_t4(p_):
            return t4(sub[np.asarray(p_)])
        sub_t4.dim = v.shape[1]
        vo.check_topk_parity_large(ref3[0][sub], [pos_of[i] for i, _ in got], [s for _, s in got], 10, 0.0, margin=64, referee=sub_t4)


def test_f32_threshold_rule():
    assert float(vo.f32_threshold(0.85)) == float(np.float32(0.85))
    assert float(vo.f32_threshold(np.float64(0.85))) > 0.85  # f64 compare == next f32 up
    assert float(vo.f32_threshold(np.float64(0.5))) == 0.5
    from typeagent_py_amd._native import f32_threshold

    for x in [0.0, 0.85, 0.7, 1.0, 1.5, -1.0, np.float64(0.85), np.float64(0.3), np.float32(0.85), 1, 0]:
        assert float(f32_threshold(x)) == float(vo.f32_threshold(x))


def test_chunked_oracle_equals_whole():
    v, q = make_corpus(5000, 96, 11)
    whole = vo.lookup(v, q, 32, 0.0)
    parts = vo.lookup_chunked([v[:1500], v[1500:3100], v[3100:]], q, 32, 0.0)
    assert [i for i, _ in whole] == [i for i, _ in parts]
    assert [s for _, s in whole] == [s for _, s in parts]


def test_l2_normalize_rows():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((50, 33)).astype(np.float32)
    x[7] = 0
    y = vo.l2_normalize_rows(x)
    assert y.dtype == np.float32
    assert np.all(y[7] == 0)
    n = np.linalg.norm(np.delete(y, 7, axis=0), axis=1)
    assert np.all(np.abs(n - 1) < 1e-6)  # reference tests/test_embeddings.py:112-120
