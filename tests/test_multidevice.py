"""One VectorBase over several devices from one process (typeagent_py_amd/multidevice.py).

CPU part: the host logic of `DeviceGroup` (shard layout, appends, re-sharding, key merging, subset cursors) against the
oracle, with a numpy stand-in for the per-device engine (test infrastructure: it computes a shard's answer with the
oracle and packs it into result keys exactly like the kernels do).
GPU part (`-m gpu`): the real thing, several contexts on the visible GPU(s), through the drop-in VectorBase API.
"""

import numpy as np
import pytest

from oracle import vectorbase_oracle as vo
from tests.fake_engine import FakeEngine
from tests.fakes import NullModel
from tests.synth import make_corpus, make_queries, subset_choice
from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase, _native, multidevice


@pytest.fixture
def fake_group(monkeypatch):
    FakeEngine.instances = []
    monkeypatch.setattr(multidevice._native, "Engine", FakeEngine)
    yield
    FakeEngine.instances = []


def _vb(devices):
    return VectorBase(TextEmbeddingIndexSettings(NullModel()), devices=devices)


def _check(vb, v, q, k, ms):
    res = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=ms)
    rep = vo.check_topk_parity(vo.scores_full(v, q), [r.item for r in res], [r.score for r in res], k, ms)
    assert rep.ordinals_bit_exact
    return res


def test_group_lookup_batch_and_layout_on_fake_devices(fake_group):
    v, q = make_corpus(1003, 48, 11)
    vb = _vb([0, 1, 2])
    vb.add_embeddings(None, v)
    _check(vb, v, q, 32, 0.0)
    g = vb.engine
    assert g.bounds == [0, 335, 670, 1003] and [e.ordinal_base for e in g.engines] == [0, 335, 670]
    qs = make_queries(9, 48, 12)
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=10, min_score=0.5)
    for qi in range(9):
        ref = vo.lookup(v, qs[qi], 10, 0.5)
        assert [r.item for r in out[qi]] == [i for i, _ in ref]
    # ties across shards resolve to the smaller global ordinal
    v2 = np.concatenate([v, v[:5]])
    vb2 = _vb([0, 1])
    vb2.add_embeddings(None, v2)
    res = vb2.fuzzy_lookup_embedding(v[3], max_hits=2, min_score=0.0)
    assert [r.item for r in res] == [3, 1006]


def test_group_appends_go_to_the_last_shard_until_it_doubles(fake_group):
    v, q = make_corpus(900, 32, 21)
    vb = _vb([0, 1, 2])
    vb.add_embeddings(None, v[:300])
    _check(vb, v[:300], q, 10, 0.0)
    e0, e1, e2 = vb.engine.engines
    n0 = (len(e0.uploads), len(e1.uploads))
    vb.add_embeddings(None, v[300:350])  # fits the last shard: only it is touched
    _check(vb, v[:350], q, 10, 0.0)
    assert (len(e0.uploads), len(e1.uploads)) == n0 and e2.uploads[-1] == (100, 50)
    vb.add_embeddings(None, v[350:])  # 900 rows > 4 blocks of 100: re-shard from scratch
    _check(vb, v, q, 10, 0.0)
    assert vb.engine.bounds == [0, 300, 600, 900]
    vb.add_embedding(None, v[0])
    assert _check(vb, np.concatenate([v, v[:1]]), v[0], 2, 0.0)[0].item == 0


def test_group_subset_paging_and_predicate_on_fake_devices(fake_group):
    v, q = make_corpus(700, 32, 31)
    vb = _vb([0, 1, 2, 3])
    vb.add_embeddings(None, v)
    sub = subset_choice(700, 200, 32) + [5, 5, -3]
    res = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=20, min_score=0.0)
    ref = vo.lookup_in_subset(v, q, sub, 20, 0.0)
    assert [(r.item, r.score) for r in res] == [(i, pytest.approx(s, abs=1e-6)) for i, s in ref]
    # paged paths: more hits than one page, all survivors, predicate
    big = vb.fuzzy_lookup_embedding(q, max_hits=600, min_score=0.0)
    ref = vo.lookup(v, q, 600, 0.0)
    assert [r.item for r in big] == [i for i, _ in ref]
    allhits = vb.fuzzy_lookup_embedding(q, max_hits=0, min_score=0.5)
    assert [r.item for r in allhits] == [i for i, _ in vo.lookup(v, q, 0, 0.5)]
    pred = vb.fuzzy_lookup_embedding(q, max_hits=7, min_score=0.0, predicate=lambda i: i % 3 == 0)
    assert [r.item for r in pred] == [i for i, _ in vo.lookup(v, q, 7, 0.0, predicate=lambda i: i % 3 == 0)]
    sub_all = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=0, min_score=0.0)  # paged subset search across shards
    ref_all = vo.lookup_in_subset(v, q, sub, 0, 0.0)
    assert sorted((r.item, round(r.score, 6)) for r in sub_all) == sorted((i, round(s, 6)) for i, s in ref_all)
    assert [r.score for r in sub_all] == sorted((r.score for r in sub_all), reverse=True)


def test_merge_keys_host_helper():
    rng = np.random.default_rng(5)
    lists = np.sort(rng.integers(1, 1 << 62, size=(5, 7, 9), dtype=np.uint64), axis=2)[:, :, ::-1].copy()
    lists[2, 3, 4:] = 0  # a short list
    got = _native.merge_keys(lists)
    for q in range(7):
        want = np.sort(lists[:, q, :].reshape(-1))[::-1][:9]
        np.testing.assert_array_equal(got[q], want)


def test_merge_keys_host_random_shapes():
    """tavb_merge_keys_host (the merge of a one-launch small lookup's per-workgroup lists, of several devices' lists, of begin/end pages) against a
    plain sort: more lists than k and fewer, lists that are short or empty, exact duplicates across lists, one list, k = 1."""
    rng = np.random.default_rng(77)
    shapes = [(204, 1, 10), (40, 1, 50), (64, 1, 32), (8, 1, 256), (1, 3, 5), (300, 2, 1), (3, 2, 64), (17, 4, 17)]
    shapes += [(8, 1, 50), (16, 1, 10), (2, 1, 64), (8, 1, 1), (16, 1, 64), (1, 1, 7)]  # a few lists, one query: the plain k-way merge (grouped one-launch lookups)
    shapes += [(int(rng.integers(1, 17)), 1, int(rng.integers(1, 70))) for _ in range(60)]
    shapes += [(int(rng.integers(1, 260)), int(rng.integers(1, 4)), int(rng.integers(1, 70))) for _ in range(200)]
    for trial, (n_lists, nq, k) in enumerate(shapes):
        keys = rng.integers(1, (1 << 63) - 1, size=(n_lists, nq, k), dtype=np.int64).astype(np.uint64)
        if trial % 3 == 0:
            keys = keys % np.uint64(40) + np.uint64(1)  # many duplicates
        keys = np.where(rng.random((n_lists, nq, k)) < rng.random(), keys, np.uint64(0))  # holes: sorted to the tails below
        lists = np.sort(keys, axis=2)[:, :, ::-1].copy()
        got = _native.merge_keys(lists)
        for q in range(nq):
            flat = np.sort(lists[:, q, :].reshape(-1))[::-1]
            want = np.zeros(k, dtype=np.uint64)
            want[: min(k, flat.size)] = flat[:k]
            np.testing.assert_array_equal(got[q], want, err_msg=f"{n_lists} lists, k={k}, query {q}")


# ------------------------------------------------------------------------------------------------------------------
# GPU: several contexts on the visible device(s)
# ------------------------------------------------------------------------------------------------------------------
def _device_list(n):
    have = _native.device_count()
    return [i % max(have, 1) for i in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_gpu_device_group_matches_oracle_through_the_vectorbase_api(dtype):
    v, q = make_corpus(50_021, 1536, 41)
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), devices=_device_list(3), corpus_dtype=dtype)
    vb.add_embeddings(None, v)
    vv = v.astype(np.float16).astype(np.float32) if dtype == "fp16" else v
    res = vb.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    assert vo.check_topk_parity(vo.scores_full(vv, q), [r.item for r in res], [r.score for r in res], 32, 0.0).ordinals_bit_exact
    assert max(r.item for r in res) > 16_674  # hits from the later shards carry global ordinals
    for nq in (7, 40, 300):  # streaming tier, 64-query tile, 256-query tile + rescoring: per shard, merged on the host
        qs = make_queries(nq, 1536, 42 + nq)
        out = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
        for qi in range(0, nq, max(1, nq // 12)):
            rep = vo.check_topk_parity(vo.scores_full(vv, qs[qi]), [r.item for r in out[qi]], [r.score for r in out[qi]], 32, 0.0)
            assert rep.tie_permuted_positions <= 2  # only inside fp32 near-tie groups of the reference (the checker verified that)
    sub = subset_choice(50_021, 3000, 43) + [7, 7, -1]
    res = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=25, min_score=0.0)
    ref = vo.lookup_in_subset(vv, q, sub, 25, 0.0)
    assert [r.item for r in res] == [i for i, _ in ref]
    big = vb.fuzzy_lookup_embedding(q, max_hits=700, min_score=0.0)  # paged across shards
    assert [r.item for r in big] == [i for i, _ in vo.lookup(vv, q, 700, 0.0)]
    pred = vb.fuzzy_lookup_embedding(q, max_hits=9, min_score=0.0, predicate=lambda i: i % 5 == 1)
    assert [r.item for r in pred] == [i for i, _ in vo.lookup(vv, q, 9, 0.0, predicate=lambda i: i % 5 == 1)]
    vb.add_embeddings(None, v[:100])  # append into the last shard
    res = vb.fuzzy_lookup_embedding(v[5], max_hits=2, min_score=0.0)
    assert [r.item for r in res] == [5, 50_026]


@pytest.mark.gpu
def test_gpu_device_group_adopts_one_tensor_per_device():
    import torch

    v, q = make_corpus(9_001, 1536, 51)
    devs = _device_list(2)
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), devices=devs)
    parts = [torch.from_numpy(v[:4000]).to(f"cuda:{devs[0]}"), torch.from_numpy(v[4000:]).to(f"cuda:{devs[1]}")]
    vb.adopt_device_corpus(parts)
    assert len(vb) == 9_001
    res = vb.fuzzy_lookup_embedding(q, max_hits=32, min_score=0.0)
    assert vo.check_topk_parity(vo.scores_full(v, q), [r.item for r in res], [r.score for r in res], 32, 0.0).ordinals_bit_exact
    np.testing.assert_array_equal(vb.serialize(), v)


@pytest.mark.gpu
def test_gpu_two_real_devices_when_present():
    if _native.device_count() < 2:
        pytest.skip("one GPU visible")
    v, q = make_corpus(200_003, 1536, 61)
    vb = VectorBase(TextEmbeddingIndexSettings(NullModel()), devices=[0, 1], corpus_dtype="fp16")
    vb.add_embeddings(None, v)
    vv = v.astype(np.float16).astype(np.float32)
    qs = make_queries(130, 1536, 62)
    out = vb.fuzzy_lookup_embeddings(qs, max_hits=32, min_score=0.0)
    for qi in range(0, 130, 13):
        assert vo.check_topk_parity(vo.scores_full(vv, qs[qi]), [r.item for r in out[qi]], [r.score for r in out[qi]], 32, 0.0).ordinals_bit_exact


def _rccl_rank(rank: int, world: int, id_path: str, n: int, ret) -> None:
    """One process per GPU: shard of the seeded corpus on cuda:<rank>, libtavb's own RCCL communicator (rendezvous id through a
    file: no torch.distributed anywhere), collective lookups."""
    import os
    import time

    try:
        import torch

        from typeagent_py_amd.sharded import DeviceShardBackend, ShardedSearcher, shard_range

        v, _ = make_corpus(n, 1536, 71)
        lo, hi = shard_range(n, world, rank)
        backend = DeviceShardBackend(rank)
        with torch.cuda.stream(backend.stream):
            shard = torch.from_numpy(v[lo:hi]).to(f"cuda:{rank}").half()
        backend.set_shard(shard, row_offset=lo)

        def exchange(uid):
            if uid is not None:
                with open(id_path + ".tmp", "wb") as f:
                    f.write(uid)
                os.replace(id_path + ".tmp", id_path)
                return uid
            for _ in range(600):
                if os.path.exists(id_path):
                    return open(id_path, "rb").read()
                time.sleep(0.05)
            raise TimeoutError("no rendezvous id")

        backend.init_comm(rank, world, exchange_id=exchange)
        searcher = ShardedSearcher(backend)
        out = {}
        for nq in (1, 40, 130):
            qs = make_queries(nq, 1536, 72 + nq)
            res = searcher.search(torch.from_numpy(qs).to(f"cuda:{rank}"), 32, 0.0)
            out[nq] = (res.ordinals.copy(), res.scores.copy(), res.counts.copy())
        # subset and predicate forms over the two shards (tavb_search_subset_device + remap + tavb_allgather_merge)
        from typeagent_py_amd.sharded import ShardedVectorBase

        svb = ShardedVectorBase(backend, lo, hi - lo, n)
        qs = make_queries(4, 1536, 99)
        subset = np.random.default_rng(98).integers(-100, n, size=5000).tolist()
        out["subset"] = [(r.item, r.score) for r in svb.fuzzy_lookup_embedding_in_subset(qs[0], subset, max_hits=32, min_score=0.0)]
        out["pred"] = [(r.item, r.score) for r in svb.fuzzy_lookup_embedding(qs[1], max_hits=16, min_score=0.52, predicate=lambda i: i % 5 == 2)]
        ret[rank] = out
    except Exception as exc:  # noqa: BLE001
        import traceback

        ret[rank] = "ERROR: " + "".join(traceback.format_exception(exc))


@pytest.mark.gpu
def test_gpu_two_ranks_rccl_through_the_c_abi_when_two_gpus_are_present(tmp_path):
    """N = 2 on real hardware wherever a second GPU shows up: two processes, one per GPU, the exchange = `tavb_search_allgather`
    (ncclAllGather issued by libtavb on its own stream).  Every rank must return the whole-corpus answer."""
    if _native.device_count() < 2:
        pytest.skip("one GPU visible")
    import torch.multiprocessing as mp

    n = 300_001
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_rccl_rank, args=(r, 2, str(tmp_path / "rccl_id"), n, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(not p.is_alive() and p.exitcode == 0 for p in procs)
    assert not isinstance(ret[0], str), ret[0]
    assert not isinstance(ret[1], str), ret[1]
    v, _ = make_corpus(n, 1536, 71)
    vv = v.astype(np.float16).astype(np.float32)
    assert ret[0]["subset"] == ret[1]["subset"] and ret[0]["pred"] == ret[1]["pred"]
    qs4 = make_queries(4, 1536, 99)
    subset = np.random.default_rng(98).integers(-100, n, size=5000)
    vo.check_topk_parity(vo.scores_full(vv, qs4[0])[subset], [i for i, _ in ret[0]["subset"]], [s_ for _, s_ in ret[0]["subset"]], 32, 0.0, candidate_ordinals=subset)
    want = vo.lookup(vv, qs4[1], 16, 0.52, predicate=lambda i: i % 5 == 2)
    assert [i for i, _ in ret[0]["pred"]] == [i for i, _ in want]
    for nq in (1, 40, 130):
        for a, b in zip(ret[0][nq], ret[1][nq]):
            np.testing.assert_array_equal(a, b)  # both ranks: the same answer
        qs = make_queries(nq, 1536, 72 + nq)
        ords, scs, cnts = ret[0][nq]
        for qi in range(0, nq, max(1, nq // 10)):
            m = int(cnts[qi])
            assert m == 32
            vo.check_topk_parity(vo.scores_full(vv, qs[qi]), ords[qi, :m].tolist(), scs[qi, :m].tolist(), 32, 0.0)


# ------------------------------------------------------------------------------------------------------------------
# seeded random histories on fake devices: appends of random sizes interleaved with every kind of lookup
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(40))
def test_group_random_history_on_fake_devices(fake_group, seed):
    rng = np.random.default_rng(5000 + seed)
    n_dev = int(rng.integers(1, 6))
    dim = int(rng.choice([3, 16, 48]))
    vb = _vb(list(range(n_dev)))
    rows = np.zeros((0, dim), dtype=np.float32)
    for step in range(int(rng.integers(2, 7))):
        add = int(rng.choice([1, 2, 7, 40, 300]))
        fresh = rng.standard_normal((add, dim)).astype(np.float32)
        fresh /= np.linalg.norm(fresh, axis=1, keepdims=True)
        if rng.random() < 0.3 and len(rows):
            fresh[0] = rows[int(rng.integers(len(rows)))]  # an exact duplicate: ties across shards resolve to the smaller ordinal
        if add == 1:
            vb.add_embedding(None, fresh[0])
        else:
            vb.add_embeddings(None, fresh)
        rows = np.concatenate([rows, fresh])
        n = len(rows)
        assert len(vb) == n and vb.engine.bounds[-1] == n and sorted(vb.engine.bounds) == vb.engine.bounds
        q = rows[int(rng.integers(n))] if rng.random() < 0.5 else fresh[-1]
        k = int(rng.choice([1, 3, 10, 32]))
        ms = float(rng.choice([0.0, 0.4, 0.55]))
        # single
        # (exact ties -- the duplicated rows -- have no defined order in the reference; ours is ascending ordinal, also across shards)
        res = vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=ms)
        vo.check_topk_parity(vo.scores_full(rows, q), [r.item for r in res], [r.score for r in res], k, ms)
        assert all(a.score > b.score or (a.score == b.score and a.item < b.item) for a, b in zip(res, res[1:]))
        # batch
        qs = rows[rng.integers(0, n, size=int(rng.integers(2, 6)))]
        out = vb.fuzzy_lookup_embeddings(qs, max_hits=k, min_score=ms)
        for qi in range(len(qs)):
            vo.check_topk_parity(vo.scores_full(rows, qs[qi]), [r.item for r in out[qi]], [r.score for r in out[qi]], k, ms)
            assert all(a.score > b.score or (a.score == b.score and a.item < b.item) for a, b in zip(out[qi], out[qi][1:]))
        # subset (duplicates and negative ordinals allowed), also with more hits than one page
        sub = rng.integers(-min(n, 3), n, size=int(rng.integers(1, min(n, 50) + 1))).tolist()
        got = vb.fuzzy_lookup_embedding_in_subset(q, sub, max_hits=k, min_score=ms)
        sub_a = np.asarray(sub, dtype=np.int64)
        vo.check_topk_parity(vo.scores_full(rows, q)[sub_a], [r.item for r in got], [r.score for r in got], k, ms, candidate_ordinals=sub_a)
        assert len(got) == len(vo.lookup_in_subset(rows, q, sub, k, ms))
        # all survivors / predicate
        everything = vb.fuzzy_lookup_embedding(q, max_hits=0, min_score=0.5)
        assert sorted(r.item for r in everything) == sorted(i for i, _ in vo.lookup(rows, q, 0, 0.5))
        assert [r.score for r in everything] == sorted((r.score for r in everything), reverse=True)
        pred = vb.fuzzy_lookup_embedding(q, max_hits=5, min_score=0.0, predicate=lambda i: i % 2 == 0)
        want = vo.lookup(rows, q, 5, 0.0, predicate=lambda i: i % 2 == 0)  # (the predicate branch IS stable: ties by ascending ordinal, vectorbase.py:200)
        assert [r.item for r in pred] == [i for i, _ in want]
    np.testing.assert_array_equal(vb.serialize(), rows)


def test_group_growing_from_a_small_seed_stays_balanced(fake_group):
    """Layout policy of a device group that is grown by appends.  Rounds 2 and 3 pulled in opposite directions: round 2's advice counted
    re-shards (a block of ceil(n / g) re-shards ~g ln N times), round 3's measured what doubling the block did to lookups (right after a
    re-shard 5 of 8 devices held rows, ~1.8x slower until the index had doubled).  Lookups are the hot path: every (re-)shard lays the
    rows out balanced, every device reserves twice its block, appends go to the last shard until it outgrows that.  So: all devices hold
    rows after every re-shard, no shard is ever more than twice the balanced size, and the re-shards stay within ~g ln(N / n0)."""
    d, g = 8, 8
    v, q = make_corpus(6000, d, 41)
    vb = _vb(list(range(g)))
    vb.add_embeddings(None, v[:16])
    vb.fuzzy_lookup_embedding(q, max_hits=1)
    full_uploads = moved = 0
    e0 = vb.engine.engines[0]
    seen = len(e0.uploads)
    n = 16
    worst_skew = 0.0
    while n < 6000:
        step = min(37, 6000 - n)
        vb.add_embeddings(None, v[n : n + step])
        n += step
        vb.fuzzy_lookup_embedding(q, max_hits=1)
        b = vb.engine.bounds
        sizes = [hi - lo for lo, hi in zip(b, b[1:])]
        worst_skew = max(worst_skew, max(sizes) / (n / g))
        if len(e0.uploads) > seen:  # shard 0 is only ever written by a (re-)shard from row 0
            full_uploads += len(e0.uploads) - seen
            moved += n
            seen = len(e0.uploads)
            assert min(sizes) > 0 or n < g, sizes                      # every device holds rows right after a re-shard ...
            assert max(sizes) <= -(-n // g), (sizes, n)                # ... and the layout is the balanced one
    assert worst_skew <= 2.0, worst_skew                               # the last shard never holds more than twice its balanced share
    assert full_uploads <= g * np.log(6000 / 16) + g, full_uploads     # ~g ln(N / n0)
    assert moved <= (g + 2) * 6000, moved
    _check(vb, v, q, 10, 0.0)


def test_group_lookups_from_several_threads_do_not_mix_their_results(fake_group):
    import threading

    v, _ = make_corpus(2000, 16, 51)
    qs = make_queries(40, 16, 52)
    vb = _vb([0, 1, 2])
    vb.add_embeddings(None, v)
    want = [[i for i, _ in vo.lookup(v, q, 5, 0.0)] for q in qs]
    errors = []

    def worker(lo):
        try:
            for rep in range(20):
                for qi in range(lo, 40, 4):
                    if rep % 2:
                        got = vb.fuzzy_lookup_embedding(qs[qi], max_hits=5, min_score=0.0)
                    else:
                        got = vb.fuzzy_lookup_embeddings(qs[qi : qi + 1], max_hits=5, min_score=0.0)[0]
                    assert [r.item for r in got] == want[qi], qi
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:1]
