"""CPU suite: the N > 1 path (row-sharded corpus, all-gather of per-shard top-k keys,
merge) with world_size = 2 over gloo.  The communication pattern, shard ranges, ordinal
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
    assert set(ret.keys()) == {0, 1}
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
