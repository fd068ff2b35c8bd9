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
