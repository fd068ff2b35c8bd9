"""GPU suite: seeded random differential cases through the drop-in class -- shapes, dtypes, batch sizes, k, thresholds, subsets and
degenerate rows (duplicates = exact ties, zero rows, un-normalised rows, NaN / inf rows) drawn at random, every answer checked with the
tie-aware parity checker against the numpy oracle.  The hand-picked cases live in tests/test_gpu_parity.py; this file is there for the
combinations nobody thought of.  Nothing here reads /root/reference."""

import os

import numpy as np
import pytest

from oracle import vectorbase_oracle as vo
from tests.fakes import NullModel
from tests.synth import make_clustered_corpus
from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase

pytestmark = pytest.mark.gpu

DIMS = [1, 2, 3, 7, 16, 33, 64, 96, 100, 128, 384, 768, 1000, 1024, 1536, 2048, 3072]  # (2048 / 3072 since round 6: the case list of a seed base changed with them)
BATCHES = [1, 1, 2, 3, 5, 8, 9, 31, 32, 33, 64, 65, 100, 129, 257]
KS = [1, 2, 5, 10, 32, 48, 49, 64, 65, 100, 300]
THRESHOLDS = [None, 0.0, 0.3, 0.5, 0.52, 0.6, 0.85, 1.0, 1.5, -0.2]


def _case(seed: int):
    rng = np.random.default_rng(seed)
    d = int(rng.choice(DIMS))
    n = int(rng.choice([1, 2, 17, 64, 255, 256, 257, 1000, 4099, 20_000])) if d >= 64 else int(rng.choice([1, 5, 300, 3000]))
    nq = int(rng.choice(BATCHES))
    k = int(rng.choice(KS))
    ms = THRESHOLDS[int(rng.integers(len(THRESHOLDS)))]
    dtype = "fp16" if rng.random() < 0.5 else "fp32"
    v = rng.standard_normal((n, d)).astype(np.float32)
    norms = np.linalg.norm(v, axis=1, keepdims=True)
    v /= np.where(norms == 0, 1, norms)
    flavour = int(rng.integers(6))
    if flavour == 1 and n > 4:  # exact ties: duplicated rows
        src = rng.integers(0, n, size=max(1, n // 5))
        dst = rng.integers(0, n, size=src.size)
        v[dst] = v[src]
    elif flavour == 2 and n > 2:  # zero rows score exactly 0.5
        v[rng.integers(0, n, size=max(1, n // 10))] = 0.0
    elif flavour == 3:  # un-normalised rows: dot products outside [-1, 1] clip to 0 / 1
        v *= rng.uniform(0.1, 4.0, size=(n, 1)).astype(np.float32)
    elif flavour == 4 and n > 3 and dtype == "fp32":  # NaN / inf rows never pass `>=` (inf - inf = NaN in the dot)
        v[int(rng.integers(n)), 0] = np.nan
        v[int(rng.integers(n)), d - 1] = np.inf
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    if n > 1:
        q[0] = v[int(rng.integers(n))]  # a planted match (possibly a zero / NaN row: then it is just another query)
        if not np.all(np.isfinite(q[0])) or not np.any(q[0]):
            q[0] = q[-1]
    subset = None
    if rng.random() < 0.25 and n > 3:
        subset = rng.integers(0, n, size=int(rng.integers(1, min(n, 500)))).tolist()  # duplicates allowed
    return dict(d=d, n=n, nq=nq, k=k, ms=ms, dtype=dtype, v=v, q=q, subset=subset, flavour=flavour)


FUZZ_BASE = int(os.environ.get("TAVB_FUZZ_BASE", "1000"))  # another 250 cases: TAVB_FUZZ_BASE=2000 pytest tests/test_gpu_fuzz.py -m gpu


@pytest.mark.parametrize("seed", range(250))
def test_random_case_against_the_oracle(seed):
    c = _case(FUZZ_BASE + seed)
    v, q, k, ms = c["v"], c["q"], c["k"], c["ms"]
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=c["dtype"])
    vb.add_embeddings(None, v)
    seen = v.astype(np.float16).astype(np.float32) if c["dtype"] == "fp16" else v  # the values the kernels multiply
    ms_eff = 0.0 if ms is None else ms
    tag = {key: c[key] for key in ("d", "n", "nq", "k", "ms", "dtype", "flavour")}
    if c["subset"] is not None:
        for qi in range(min(c["nq"], 3)):
            got = vb.fuzzy_lookup_embedding_in_subset(q[qi], c["subset"], max_hits=k, min_score=ms)
            sub = np.asarray(c["subset"], dtype=np.int64)
            ref = vo.scores_full(seen, q[qi])[sub]
            vo.check_topk_parity(ref, [r.item for r in got], [r.score for r in got], k, ms_eff, candidate_ordinals=sub, referee=vo.f64_referee(seen[sub], q[qi]))
        return
    if c["nq"] == 1:
        batches = [vb.fuzzy_lookup_embedding(q[0], max_hits=k, min_score=ms)]
    else:
        batches = vb.fuzzy_lookup_embeddings(q, max_hits=k, min_score=ms)
    assert len(batches) == c["nq"], tag
    step = max(1, c["nq"] // 12)
    for qi in list(range(0, c["nq"], step)) + [c["nq"] - 1]:
        got = batches[qi]
        ref = vo.scores_full(seen, q[qi])
        try:
            vo.check_topk_parity(ref, [r.item for r in got], [r.score for r in got], k, ms_eff, referee=vo.f64_referee(seen, q[qi]))
        except AssertionError as exc:
            raise AssertionError(f"{tag} query {qi} tier {vb.engine.get_option('last_tier')}: {exc}") from exc
        assert all(0.0 <= r.score <= 1.0 for r in got), tag


@pytest.mark.parametrize("seed", range(80))
def test_random_grouped_batch_is_the_sequential_lookups_bit_for_bit(seed):
    """The grouped one-launch form (batches of 2 .. 128 queries on small corpora, end of round 6) with a FORCED group size -- every scan kernel
    family (1536-wide register tier, 16-byte vector tier, element tier of odd widths), both dtypes, random k <= 64 and thresholds, degenerate
    rows: each query's answer is its single lookup's, item for item and bit for bit (same per-row arithmetic and order), through the
    host-synchronous call and through the device-resident one."""
    import torch

    from typeagent_py_amd import _native

    c = _case(FUZZ_BASE + 5000 + seed)
    rng = np.random.default_rng(FUZZ_BASE + 9000 + seed)
    nq = int(rng.choice([2, 3, 5, 8, 9, 17, 32, 33, 64, 65, 100, 128]))
    k = int(rng.choice([1, 2, 5, 10, 32, 50, 64]))
    if nq > 64:
        k = min(k, 32)  # (the lists of one launch hold 32768 keys: eight row workgroups x 128 queries x 32)
    group = int(rng.choice([1, 2, 4, 8]))
    v, ms = c["v"], c["ms"]
    q = rng.standard_normal((nq, c["d"])).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = c["q"][0]
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=c["dtype"])
    vb.add_embeddings(None, v)
    eng = vb.engine
    tag = dict(d=c["d"], n=c["n"], nq=nq, k=k, ms=ms, dtype=c["dtype"], flavour=c["flavour"], group=group)
    singles = [[(r.item, r.score) for r in vb.fuzzy_lookup_embedding(q[qi], max_hits=k, min_score=ms)] for qi in range(nq)]
    eng.set_option("direct_group", group)
    out = vb.fuzzy_lookup_embeddings(q, max_hits=k, min_score=ms)
    small = v.shape[0] * v.shape[1] * (2 if c["dtype"] == "fp16" else 4) <= 128 << 20  # (`small_direct_bytes`: bigger corpora keep the tiles)
    assert eng.get_option("last_direct") == (3 if small else 0), tag
    for qi in range(nq):
        assert [(r.item, r.score) for r in out[qi]] == singles[qi] or not small, (tag, qi)
    keys = eng.search_device(torch.from_numpy(q).cuda(), k, float(_native.f32_threshold(0.0 if ms is None else ms)))
    eng.synchronize()
    assert eng.get_option("last_direct") == (4 if small else 0), tag
    if not small:
        return
    ords, scs, cnts = _native.decode_keys(keys.cpu().numpy())
    for qi in range(nq):
        m = int(cnts[qi])
        assert list(zip(ords[qi, :m].tolist(), scs[qi, :m].tolist())) == singles[qi], (tag, qi, "device-resident form")
    seen = v.astype(np.float16).astype(np.float32) if c["dtype"] == "fp16" else v
    vo.check_topk_parity(vo.scores_full(seen, q[0]), [r.item for r in out[0]], [r.score for r in out[0]], k, 0.0 if ms is None else ms,
                         referee=vo.f64_referee(seen, q[0]))


@pytest.mark.parametrize("cluster_rows,nq,k,expect_flagged", [(40, 130, 32, False), (100, 300, 32, False), (100, 1024, 50, False), (300, 130, 10, False),
                                                               (1500, 130, 32, False), (2500, 130, 32, True)])
def test_clustered_corpus_batches_against_the_oracle(cluster_rows, nq, k, expect_flagged):
    """The mid-size twin of bench.py's cfg3_clustered: every query sits next to a cluster of near-duplicate rows (incl. exact duplicates),
    so the scores around rank k are packed far inside the fp16 filter's error bound.  The wide tile keeps the whole BAND below the k-th
    best (a cluster, not a fixed 64 candidates) and rescoring makes the answer exact: no query may need the exact-tile fallback until a
    cluster outgrows the band capacity (2048 rows since round 6: 1500-row clusters fit, 2500-row clusters are flagged and re-run exactly, same answers)."""
    n = 60_000
    v, q, cl, qc = make_clustered_corpus(n, 1536, 4100 + cluster_rows, cluster_rows=cluster_rows, n_queries=nq)
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype="fp16")
    vb.add_embeddings(None, v)
    out = vb.fuzzy_lookup_embeddings(q, max_hits=k, min_score=0.0)
    assert vb.engine.get_option("last_tier") == 4
    flagged = vb.engine.get_option("last_flagged")
    assert (flagged > 0) == expect_flagged, flagged
    seen = v.astype(np.float16).astype(np.float32)
    for qi in sorted(set(np.linspace(0, nq - 1, 20).astype(int).tolist())):
        ref = vo.scores_full(seen, q[qi])
        items = [r.item for r in out[qi]]
        vo.check_topk_parity(ref, items, [r.score for r in out[qi]], k, 0.0, referee=vo.f64_referee(seen, q[qi]))
        members = np.flatnonzero(cl == qc[qi])
        assert set(items[: min(k, len(members))]) <= set(members.tolist())  # the best hits are the query's own cluster
        # the single-query kernel: the same answer up to fp32 near-ties (its summation order differs; rows ~1e-7 apart may swap)
        seq = vb.fuzzy_lookup_embedding(q[qi], max_hits=k, min_score=0.0)
        vo.check_topk_parity(ref, [r.item for r in seq], [r.score for r in seq], k, 0.0, referee=vo.f64_referee(seen, q[qi]))
        np.testing.assert_allclose([r.score for r in out[qi]], [r.score for r in seq], atol=1e-6, rtol=0)


@pytest.mark.parametrize("dtype,nq,k,ms", [("fp16", 130, 32, 0.0), ("fp16", 1024, 10, 0.9), ("fp32", 300, 50, 0.0)])
def test_anisotropic_corpus_batches_against_the_oracle(dtype, nq, k, ms):
    """Real embedding corpora are not isotropic: all rows share a large common component, so every pairwise cosine sits in a narrow range
    near 0.8 (score 0.9) and the scores around rank k are an order of magnitude denser than on gaussian data.  The band below the k-th
    best then holds more rows than k + 2, but nowhere near its capacity: no fallback, exact answers (fp32 corpora: through the fp16
    shadow + fp32 rescoring)."""
    rng = np.random.default_rng(5200 + nq)
    n, d = 50_000, 1536
    common = rng.standard_normal(d).astype(np.float32)
    common /= np.linalg.norm(common)
    v = rng.standard_normal((n, d)).astype(np.float32) / np.float32(np.sqrt(d))  # unit-ish noise
    v = 2.0 * common[None, :] + v  # cosine between two rows ~ 4 / 5 = 0.8
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = rng.standard_normal((nq, d)).astype(np.float32) / np.float32(np.sqrt(d)) + 2.0 * common[None, :]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = v[n // 3]
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=dtype)
    vb.add_embeddings(None, v)
    out = vb.fuzzy_lookup_embeddings(q, max_hits=k, min_score=ms)
    assert vb.engine.get_option("last_tier") == 4 and vb.engine.get_option("last_flagged") == 0
    seen = v.astype(np.float16).astype(np.float32) if dtype == "fp16" else v
    for qi in sorted(set(np.linspace(0, nq - 1, 16).astype(int).tolist())):
        ref = vo.scores_full(seen, q[qi])
        assert float(np.median(ref)) > 0.85  # the whole corpus scores high: dense around rank k
        vo.check_topk_parity(ref, [r.item for r in out[qi]], [r.score for r in out[qi]], k, ms, referee=vo.f64_referee(seen, q[qi]))
    assert out[0][0].item == n // 3


@pytest.mark.parametrize("dtype,cluster_rows,nq,k", [("fp16", 100, 130, 32), ("fp32", 100, 300, 32), ("fp16", 200, 1024, 64), ("fp32", 60, 130, 10)])
def test_clustered_corpus_with_a_threshold_inside_the_cluster(dtype, cluster_rows, nq, k):
    """min_score set INSIDE the pack of near-equal scores of each query's cluster (the median score of its best cluster_rows rows): the filter
    admits down to min_score - 2 delta, the rescoring re-tests the exact score against min_score -- every returned row passes it, none that
    passes is missing, on fp16 corpora and on fp32 ones (fp16 shadow as the filter, fp32 rows for the exact scores)."""
    n = 50_000
    v, q, cl, qc = make_clustered_corpus(n, 1536, 4300 + cluster_rows + nq, cluster_rows=cluster_rows, n_queries=nq)
    seen = v.astype(np.float16).astype(np.float32) if dtype == "fp16" else v
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=dtype)
    vb.add_embeddings(None, v)
    probe = sorted(set(np.linspace(0, nq - 1, 12).astype(int).tolist()))
    thresholds = {}
    for qi in probe:
        ref = vo.scores_full(seen, q[qi])
        thresholds[qi] = float(np.sort(ref)[::-1][cluster_rows // 2])
    ms = float(np.median(list(thresholds.values())))  # one threshold for the batch, inside most clusters' packs
    out = vb.fuzzy_lookup_embeddings(q, max_hits=k, min_score=ms)
    assert vb.engine.get_option("last_tier") == 4 and vb.engine.get_option("last_flagged") == 0
    for qi in probe:
        ref = vo.scores_full(seen, q[qi])
        vo.check_topk_parity(ref, [r.item for r in out[qi]], [r.score for r in out[qi]], k, ms, referee=vo.f64_referee(seen, q[qi]))
        assert all(r.score >= np.float32(ms) for r in out[qi])


@pytest.mark.parametrize("seed", range(12))
def test_random_batches_through_the_threshold_ladder(seed):
    """The random cases above stay below 20k rows: no threshold ladder, hardly an admission after the first tile.  Here: 170k - 400k rows of
    64 - 512 dimensions, batches of 33 - 700 queries (64-query tile, 128- and 256-query tiles, fp32 corpora through the fp16 shadow), every k
    and threshold the consumers use, duplicated rows; several ladder phases, sparse admissions, rescoring -- against the oracle with the float64
    referee."""
    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([64, 128, 256, 512]))
    n = int(rng.choice([170_000, 250_000, 400_000]))
    nq = int(rng.choice([33, 64, 65, 100, 128, 129, 256, 300, 512, 700]))
    k = int(rng.choice([1, 10, 25, 32, 50, 64]))
    ms = [None, 0.0, 0.5, 0.55, 0.6][int(rng.integers(5))]
    dtype = "fp16" if rng.random() < 0.6 else "fp32"
    v = rng.standard_normal((n, d)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    src = rng.integers(0, n, size=2000)
    v[rng.integers(0, n, size=2000)] = v[src]  # exact ties
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = v[int(src[0])]
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), corpus_dtype=dtype)
    vb.add_embeddings(None, v)
    out = vb.fuzzy_lookup_embeddings(q, max_hits=k, min_score=ms)
    tag = dict(d=d, n=n, nq=nq, k=k, ms=ms, dtype=dtype, tier=vb.engine.get_option("last_tier"))
    assert tag["tier"] in (4, 5), tag
    seen = v.astype(np.float16).astype(np.float32) if dtype == "fp16" else v
    ms_eff = 0.0 if ms is None else ms
    for qi in sorted(set([0, nq - 1] + rng.integers(0, nq, size=6).tolist())):
        ref = vo.scores_full(seen, q[qi])
        try:
            vo.check_topk_parity(ref, [r.item for r in out[qi]], [r.score for r in out[qi]], k, ms_eff, referee=vo.f64_referee(seen, q[qi]))
        except AssertionError as exc:
            raise AssertionError(f"{tag} query {qi}: {exc}") from exc
    # the same batch one query at a time (streaming kernels): same hits up to float32 near-ties
    for qi in (0, nq - 1):
        one = vb.fuzzy_lookup_embedding(q[qi], max_hits=k, min_score=ms)
        assert len(one) == len(out[qi]), tag
        np.testing.assert_allclose([r.score for r in one], [r.score for r in out[qi]], atol=1e-6, rtol=0)
