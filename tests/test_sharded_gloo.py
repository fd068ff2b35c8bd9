"""CPU suite: the N > 1 path (row-sharded corpus, all-gather of per-shard top-k keys,
merge) with world_size = 2 and 3 (uneven and empty shards) over gloo.  The communication pattern, shard ranges, ordinal
offsets and key packing are the product's (typeagent_py_amd/sharded.py); only the two
compute hooks of the backend are replaced by host stand-ins, because there is no GPU in
this container (the same hooks run the HIP kernels in tests/test_gpu_parity.py and in
bench.py --gpus N)."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class HostStandInBackend:
    """Implements ShardBackend with the oracle for the local search and a numpy sort for
    the merge.  TEST ONLY (lives in tests/)."""

    def __init__(self, shard: np.ndarray, row_offset: int):
        self.shard = shard
        self.row_offset = row_offset

    @staticmethod
    def pack(score: float, ordinal: int) -> int:
        bits = int(np.float32(score).view(np.uint32))
        key = (bits << 32) | (0xFFFFFFFF - ordinal)
        return key - (1 << 64) if key >= (1 << 63) else key  # as int64

    def local_search(self, queries, k, thr):
        from oracle import vectorbase_oracle as vo

        q = queries.numpy()
        out = np.zeros((q.shape[0], k), dtype=np.int64)
        for qi in range(q.shape[0]):
            hits = vo.lookup(self.shard, q[qi], k, np.float32(thr))
            hits.sort(key=lambda t: (-t[1], t[0]))
            for j, (i, s) in enumerate(hits):
                out[qi, j] = self.pack(s, i + self.row_offset)
        return torch.from_numpy(out)

    def merge(self, gathered):
        g = gathered.numpy().view(np.uint64)  # [world, nq, k]
        world, nq, k = g.shape
        allk = np.transpose(g, (1, 0, 2)).reshape(nq, world * k)
        allk = np.sort(allk, axis=1)[:, ::-1]  # bigger key = better hit
        return torch.from_numpy(np.ascontiguousarray(allk[:, :k]).view(np.int64))

    def to_host(self, keys):
        return keys.numpy()

    def empty_gather(self, world, nq, k):
        return torch.empty((world, nq, k), dtype=torch.int64)

    def failed_lists(self, nq, k):
        return torch.full((nq, k), -1, dtype=torch.int64)  # PEER_FAILED_KEY in every slot

    # the forms beside the plain lookup (same contracts as DeviceShardBackend's)
    def local_search_subset(self, query, local_rows, positions, k, thr):
        from oracle import vectorbase_oracle as vo

        out = np.zeros((1, k), dtype=np.int64)
        if len(local_rows):
            sc = vo.scores_full(self.shard[np.asarray(local_rows)], query)
            keep = np.flatnonzero(sc >= np.float32(thr))
            order = keep[np.lexsort((np.asarray(positions)[keep], -sc[keep].astype(np.float64)))][:k]
            for j, i in enumerate(order):
                out[0, j] = self.pack(sc[i], int(positions[i]))
        return torch.from_numpy(out)

    def local_survivors(self, query, thr):
        from oracle import vectorbase_oracle as vo

        sc = vo.scores_full(self.shard, query)
        keep = np.flatnonzero(sc >= np.float32(thr))
        return keep.astype(np.int64) + self.row_offset, sc[keep]

    def keys_to_device(self, keys):
        return torch.from_numpy(np.ascontiguousarray(keys).view(np.int64))

    # storage hooks (same contracts as DeviceShardBackend's)
    def set_rows(self, rows, row_offset, dtype="fp32"):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        self.shard = rows if rows.ndim == 2 and rows.shape[1] > 0 else np.zeros((0, self.shard.shape[1]), dtype=np.float32)
        self.row_offset = int(row_offset)

    def append_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        self.shard = rows if self.shard.size == 0 else np.concatenate([self.shard, rows])

    def rows_to_host(self):
        return self.shard.copy()


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank: int, world: int, port: int, total_rows: int, dim: int, k: int, min_score: float, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.synth import make_corpus, make_queries
        from typeagent_py_amd.sharded import ShardedSearcher, shard_range

        v, _ = make_corpus(total_rows, dim, 31337)
        qs = make_queries(5, dim, 31338)
        lo, hi = shard_range(total_rows, world, rank)
        searcher = ShardedSearcher(HostStandInBackend(v[lo:hi], lo))
        assert searcher.world == world and searcher.rank == rank
        res = searcher.search(torch.from_numpy(qs), k, min_score)
        ret[rank] = (res.ordinals.copy(), res.scores.copy(), res.counts.copy())
        # the VectorBase-shaped front end over the same shards gives the same hits
        from typeagent_py_amd.sharded import ShardedVectorBase

        svb = ShardedVectorBase(HostStandInBackend(v[lo:hi], lo), lo, hi - lo, total_rows)
        assert len(svb) == total_rows and bool(svb)
        hits = svb.fuzzy_lookup_embedding(qs[0], max_hits=k, min_score=min_score)
        m = int(res.counts[0])
        assert [h.item for h in hits] == res.ordinals[0, :m].tolist()
        assert [h.score for h in hits] == [float(x) for x in res.scores[0, :m]]
        # subset form (duplicates, negative ordinals), predicate form, message aggregation: every rank gets the whole-corpus answer
        from oracle import messages_oracle as mo
        from oracle import vectorbase_oracle as vo

        rng = np.random.default_rng(5)
        subset = rng.integers(-min(3, total_rows), total_rows, size=min(60, 4 * total_rows)).tolist() + [0, 0]
        got = svb.fuzzy_lookup_embedding_in_subset(qs[1], subset, max_hits=k, min_score=min_score)
        want = vo.lookup_in_subset(v, qs[1], subset, k, min_score)
        sub_a = np.asarray(subset, dtype=np.int64)
        assert len(got) == len(want)  # (exact ties -- the same row named twice -- have no defined order in the reference: tie-aware check)
        vo.check_topk_parity(vo.scores_full(v, qs[1])[sub_a], [h.item for h in got], [h.score for h in got], k, min_score, candidate_ordinals=sub_a)
        pred = lambda i: i % 3 != 1
        got = svb.fuzzy_lookup_embedding(qs[2], max_hits=k, min_score=min_score, predicate=pred)
        want = vo.lookup(v, qs[2], k, min_score, predicate=pred)
        assert [(h.item, h.score) for h in got] == [(i, s) for i, s in want]
        row_to_msg = [i // 2 for i in range(total_rows)]
        got = svb.lookup_messages_by_embedding(qs[3], row_to_msg, max_matches=k, threshold_score=min_score, accept=range(0, total_rows, 2))
        look = lambda e, kk, t: vo.lookup(v, e, kk, t)
        want = mo.sqlite_lookup_by_embedding(look, qs[3], row_to_msg, k, min_score, list(range(0, total_rows, 2)))
        assert [(h.item, h.score) for h in got] == want
        with pytest.raises(IndexError):
            svb.fuzzy_lookup_embedding_in_subset(qs[1], [total_rows], max_hits=k)
        # storage over the shards (vectorbase.py:115-148, 268-287): an append between two lookups lands on the LAST rank, every rank's row
        # count advances, the next lookup sees the new rows under their global ordinals; serialize() hands out this rank's rows
        extra, _ = make_corpus(7, dim, 4242)
        extra[3] = qs[4]  # a planted best hit among the appended rows
        before = svb.fuzzy_lookup_embedding(qs[4], max_hits=k, min_score=min_score)
        svb.add_embeddings(None, extra[:5])
        svb.add_embedding("k", extra[5])
        svb.add_embeddings(["a"], extra[6:])
        assert len(svb) == total_rows + 7 and svb.local_rows == (hi - lo) + (7 if rank == world - 1 else 0)
        grown = np.concatenate([v, extra])
        after = svb.fuzzy_lookup_embedding(qs[4], max_hits=k, min_score=min_score)
        want = vo.lookup(grown, qs[4], k, min_score)
        assert after[0].item == total_rows + 3 and abs(after[0].score - 1.0) < 1e-6 and before[0].item != after[0].item
        vo.check_topk_parity(vo.scores_full(grown, qs[4]), [h.item for h in after], [h.score for h in after], k, min_score)
        assert len(after) == len(want)
        mine = svb.serialize()
        np.testing.assert_array_equal(mine, grown[svb.row_offset : svb.row_offset + svb.local_rows])
        with pytest.raises(ValueError):
            svb.add_embeddings(["one key"], extra[:2])
        # deserialize: every rank hands in its own rows (here: the two halves swapped in size), offsets are agreed by all-gather
        n_all = total_rows + 7
        cuts = [0, 2 * n_all // 3] + [2 * n_all // 3 + (i + 1) * (n_all - 2 * n_all // 3) // (world - 1) for i in range(world - 1)]  # rank 0 holds two thirds
        part = grown[cuts[rank] : cuts[rank + 1]]
        svb.deserialize(part)
        assert len(svb) == n_all and svb.row_offset == cuts[rank] and svb.local_rows == len(part)
        again = svb.fuzzy_lookup_embedding(qs[4], max_hits=k, min_score=min_score)
        assert [h.item for h in again] == [h.item for h in after], (rank, again, after)
        np.testing.assert_allclose([h.score for h in again], [h.score for h in after], atol=1e-6, rtol=0)  # (numpy's sgemv on other row ranges: the last bit may move)
        svb.clear()
        assert len(svb) == 0 and svb.fuzzy_lookup_embedding(qs[0], max_hits=k) == []
        ret[("storage", rank)] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total_rows,k,min_score", [(1001, 32, 0.0), (37, 10, 0.5), (3, 8, 0.0)])
def test_two_rank_sharded_search_equals_whole_corpus(total_rows, k, min_score):
    from oracle import vectorbase_oracle as vo
    from tests.synth import make_corpus, make_queries

    world, dim = 2, 48
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), total_rows, dim, k, min_score, ret), nprocs=world, join=True)
    assert {0, 1} <= set(ret.keys()) and ret[("storage", 0)] and ret[("storage", 1)]
    o0, s0, c0 = ret[0]
    o1, s1, c1 = ret[1]
    np.testing.assert_array_equal(o0[:, :1], o1[:, :1])
    v, _ = make_corpus(total_rows, dim, 31337)
    qs = make_queries(5, dim, 31338)
    for qi in range(5):
        np.testing.assert_array_equal(c0, c1)  # every rank holds the same global answer
        m = int(c0[qi])
        np.testing.assert_array_equal(o0[qi, :m], o1[qi, :m])
        np.testing.assert_array_equal(s0[qi, :m], s1[qi, :m])
        sc = vo.scores_full(v, qs[qi])
        rep = vo.check_topk_parity(sc, o0[qi, :m].tolist(), s0[qi, :m].tolist(), k, min_score)
        assert rep.ordinals_bit_exact  # same numpy arithmetic on both sides here


@pytest.mark.parametrize("total_rows,k,min_score", [(1000, 16, 0.0), (2, 4, 0.0)])
def test_three_rank_sharded_search_equals_whole_corpus(total_rows, k, min_score):
    """Uneven shards (334 / 333 / 333 rows) and a world with an EMPTY shard (2 rows over 3 ranks): every rank ends up with the whole corpus's answer."""
    from oracle import vectorbase_oracle as vo
    from tests.synth import make_corpus, make_queries

    world, dim = 3, 48
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), total_rows, dim, k, min_score, ret), nprocs=world, join=True)
    assert all(r in ret and ret[("storage", r)] for r in range(world))
    v, _ = make_corpus(total_rows, dim, 31337)
    qs = make_queries(5, dim, 31338)
    o0, s0, c0 = ret[0]
    for r in (1, 2):
        o, s, c = ret[r]
        np.testing.assert_array_equal(c0, c)
        for qi in range(5):
            m = int(c0[qi])
            np.testing.assert_array_equal(o0[qi, :m], o[qi, :m])
            np.testing.assert_array_equal(s0[qi, :m], s[qi, :m])
    for qi in range(5):
        m = int(c0[qi])
        rep = vo.check_topk_parity(vo.scores_full(v, qs[qi]), o0[qi, :m].tolist(), s0[qi, :m].tolist(), k, min_score)
        assert rep.ordinals_bit_exact


def _failing_worker(rank: int, world: int, port: int, fail_rank: int, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.synth import make_corpus, make_queries
        from typeagent_py_amd.sharded import PeerFailedError, ShardedSearcher, ShardedVectorBase, shard_range

        v, _ = make_corpus(900, 48, 31337)
        qs = make_queries(4, 48, 31338)
        lo, hi = shard_range(900, world, rank)

        class Flaky(HostStandInBackend):
            broken = False

            def local_search(self, queries, k, thr):
                if self.broken:
                    raise RuntimeError(f"injected failure of the local search on rank {rank}")
                return super().local_search(queries, k, thr)

        backend = Flaky(v[lo:hi], lo)
        searcher = ShardedSearcher(backend)
        first = searcher.search(torch.from_numpy(qs), 8, 0.0)
        backend.broken = rank == fail_rank
        try:
            searcher.search(torch.from_numpy(qs), 8, 0.0)
            outcome = "answer"
        except PeerFailedError:
            outcome = "peer"
        except RuntimeError as exc:
            outcome = "own" if "injected failure" in str(exc) else f"other: {exc}"
        backend.broken = False
        again = searcher.search(torch.from_numpy(qs), 8, 0.0)  # nobody was left behind in the all-gather: the next collective lines up
        np.testing.assert_array_equal(first.ordinals, again.ordinals)
        # imbalance / rebalance of the VectorBase-shaped front end: appends land on the last rank, a rebalance re-deals the rows
        svb = ShardedVectorBase(Flaky(v[lo:hi].copy(), lo), lo, hi - lo, 900)
        extra, _ = make_corpus(600, 48, 4242)
        svb.add_embeddings(None, extra)
        imb = svb.imbalance()
        before = svb.fuzzy_lookup_embedding(qs[1], max_hits=8)
        svb.rebalance()
        after = svb.fuzzy_lookup_embedding(qs[1], max_hits=8)
        # the forms beside the plain lookup follow the same protocol: a predicate (the caller's code) that raises on ONE rank, a subset search
        # whose local part fails there -- that rank joins the exchange with the failure key and raises its own error, the others PeerFailedError
        def outcome_of(call):
            try:
                call()
                return "answer"
            except PeerFailedError:
                return "peer"
            except RuntimeError as exc:
                return "own" if "injected failure" in str(exc) else f"other: {exc}"

        def pred(i):
            if rank == fail_rank:
                raise RuntimeError("injected failure inside the predicate")
            return i % 2 == 0

        o_pred = outcome_of(lambda: svb.fuzzy_lookup_embedding(qs[2], max_hits=8, predicate=pred))
        good = svb.fuzzy_lookup_embedding(qs[2], max_hits=8, predicate=lambda i: i % 2 == 0)  # the next collective lines up
        orig = svb.backend.local_search_subset

        def flaky_subset(*a, **kw):
            if rank == fail_rank:
                raise RuntimeError("injected failure of the subset search")
            return orig(*a, **kw)
        svb.backend.local_search_subset = flaky_subset
        sub = list(range(0, 1500, 7))
        o_sub = outcome_of(lambda: svb.fuzzy_lookup_embedding_in_subset(qs[3], sub, max_hits=8))
        svb.backend.local_search_subset = orig
        good_sub = svb.fuzzy_lookup_embedding_in_subset(qs[3], sub, max_hits=8)
        ret[rank] = (outcome, imb, svb.imbalance(), svb.local_rows, [(r.item, r.score) for r in before] == [(r.item, r.score) for r in after], len(svb),
                     o_pred, o_sub, all(r.item % 2 == 0 for r in good) and len(good) == 8, all(r.item in sub for r in good_sub) and len(good_sub) == 8)
    finally:
        dist.destroy_process_group()


def test_a_rank_whose_local_search_fails_still_joins_the_exchange_and_every_rank_gets_an_error():
    """The protocol of `tavb_search_allgather` (include/tavb.h), on the torch.distributed route: the failing rank contributes
    TAVB_KEY_PEER_FAILED lists, joins the all-gather and raises ITS error; the others find the key at the head of their merged lists and
    raise `PeerFailedError` -- nobody hangs, nobody returns an answer that misses a shard, and the next lookup works."""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_failing_worker, args=(world, _free_port(), 1, ret), nprocs=world, join=True)
    assert [ret[r][0] for r in range(world)] == ["peer", "own", "peer"]
    for r in range(world):
        _, imb, imb_after, local_rows, same, total, o_pred, o_sub, good_pred, good_sub = ret[r]
        assert abs(imb - 900 / 500) < 1e-9 and imb_after == 1.0 and local_rows == 500 and same and total == 1500
        assert good_pred and good_sub
    assert [ret[r][6] for r in range(world)] == ["peer", "own", "peer"]  # the predicate form
    assert [ret[r][7] for r in range(world)] == ["peer", "own", "peer"]  # the subset form


def test_shard_ranges_partition_the_rows():
    from typeagent_py_amd.sharded import shard_range

    for total in (0, 1, 7, 8, 9, 1000, 100_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans[:-1], spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
