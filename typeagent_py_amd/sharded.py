"""Row-sharded lookup across the GPUs of one node: one process per GPU
(`torch.distributed`, backend "nccl" == RCCL over xGMI), each rank owning a
contiguous range of corpus rows.

Why this shards: a row's score depends only on that row
and the query (vectorbase.py:176 of the reference), so the global top-k is the
top-k of the union of the per-shard top-k lists.  The only exchange step is an
all-gather of the per-rank packed result keys -- `nq * k * 8` bytes per rank
(256 KiB at nq=1024, k=32), latency-bound, far below xGMI link bandwidth --
followed by a local k-way merge kernel on every rank.  No all-reduce, no
tensor/pipeline parallelism: there is nothing else to exchange on this path.

Result keys are `(score bits << 32) | (0xFFFFFFFF - global_ordinal)`, so the
merge is a pure integer max-merge and ties resolve to the smaller global
ordinal on every rank identically (all ranks return the same answer).

The product backend (`DeviceShardBackend`) issues the exchange from inside
libtavb: `tavb_search_allgather` = scan kernels -> `ncclAllGather` (RCCL, on the
context's stream) -> merge kernel, written straight into pinned host memory.
`torch.distributed` is used ONCE, to hand rank 0's 128-byte RCCL rendezvous id to
the other ranks (`init_comm`); the lookup path does not touch it.

The compute backend is injected so that the communication pattern can be tested
on CPU with `gloo` (tests/test_sharded_gloo.py supplies a host backend: for such
backends the searcher falls back to `torch.distributed.all_gather_into_tensor`);
the product backend below is the only one this package ships and it runs the HIP
kernels.  There is no CPU fallback in the product path.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Protocol

import numpy as np

from . import _native


def shard_range(total_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous row range [lo, hi) of `rank`: the first (total % world) ranks get one extra row."""
    base, extra = divmod(total_rows, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


class ShardBackend(Protocol):
    def local_search(self, queries, k: int, thr: float):  # -> tensor int64 [nq, k] packed keys (global ordinals)
        ...

    def merge(self, gathered):  # tensor int64 [world, nq, k] -> tensor int64 [nq, k]
        ...

    def to_host(self, keys) -> np.ndarray:  # -> int64 [nq, k]
        ...

    def empty_gather(self, world: int, nq: int, k: int):  # tensor int64 [world, nq, k] on the backend's device
        ...

    def failed_lists(self, nq: int, k: int):  # tensor int64 [nq, k] of PEER_FAILED_KEY, ordered in front of the exchange that follows
        ...

    # the forms beside the plain lookup (ShardedVectorBase: subset, predicate) -------------------------------------------------------
    def local_search_subset(self, query: np.ndarray, local_rows: np.ndarray, positions: np.ndarray, k: int, thr: float):  # -> tensor int64 [1, k], keys carrying `positions`
        ...

    def local_survivors(self, query: np.ndarray, thr: float):  # -> (global ordinals int64, scores float32) of every row of the shard with score >= thr
        ...

    def keys_to_device(self, keys: np.ndarray):  # uint64 / int64 [nq, k] host keys -> tensor on the backend's device
        ...

    # storage (ShardedVectorBase.add_embedding(s) / serialize / deserialize / clear) --------------------------------------------------
    def set_rows(self, rows: np.ndarray, row_offset: int, dtype: str = "fp32") -> None:  # replace this rank's shard (float32 [n, dim]; no rows: drop it)
        ...

    def append_rows(self, rows: np.ndarray) -> None:  # append float32 [n, dim] to this rank's shard
        ...

    def rows_to_host(self) -> np.ndarray:  # this rank's rows as float32 [n, dim]
        ...


PEER_FAILED_KEY = -1  # TAVB_KEY_PEER_FAILED (all bits set) seen as int64: what a rank whose local search failed contributes to the exchange


class PeerFailedError(RuntimeError):
    """Another rank's local search failed during a collective lookup: the merged lists would be missing its shard."""


class DeviceShardBackend:
    """HIP kernels on this rank's GPU, launched on a dedicated torch stream so that
    RCCL collectives issued by torch.distributed order correctly with them."""

    def __init__(self, device: int):
        import torch

        self.torch = torch
        self.device = int(device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self.stream):
            self.engine = _native.Engine(self.device, use_torch_stream=True)
        self._pinned: dict = {}
        self._gather: dict = {}
        self._failed: dict = {}
        self._qstage: dict = {}
        self._dtype_name = "fp32"
        self.native_comm = False  # True once init_comm() has joined the library's own RCCL communicator

    def set_shard(self, tensor, row_offset: int, rows: int | None = None) -> None:
        self.engine.set_corpus_tensor(tensor, rows=rows, ordinal_base=row_offset)
        self._dtype_name = "fp16" if tensor.dtype == self.torch.float16 else "fp32"

    def storage_dtype(self) -> str:
        """"fp16" / "fp32": what this rank's shard is stored as (what `rebalance()` keeps)."""
        return self._dtype_name

    def stage_queries(self, q: np.ndarray):
        """host float32 [nq, dim] -> device tensor, through a reused pinned buffer on the backend's stream (no pageable staging copy per call)."""
        torch = self.torch
        shape = tuple(q.shape)
        slot = self._qstage.get(shape)
        if slot is None:
            if len(self._qstage) >= 8:
                self._qstage.clear()
            with torch.cuda.stream(self.stream):
                slot = self._qstage[shape] = (torch.empty(shape, dtype=torch.float32).pin_memory(),
                                              torch.empty(shape, dtype=torch.float32, device=torch.device("cuda", self.device)))
        pinned, dev = slot
        self.stream.synchronize()  # (the previous lookup that read `dev` is long done in practice; cheap)
        pinned.numpy()[...] = q
        with torch.cuda.stream(self.stream):
            dev.copy_(pinned, non_blocking=True)
        return dev

    def init_comm(self, rank: int, world: int, exchange_id=None) -> None:
        """Collective: join libtavb's own RCCL communicator (tavb_comm_init).  Rank 0 creates the rendezvous id; `exchange_id(id or None)
        -> id` distributes it (default: `torch.distributed.broadcast_object_list` over the default group, whatever its backend)."""
        uid = _native.comm_unique_id() if rank == 0 else None
        if world > 1 or exchange_id is not None:
            if exchange_id is None:
                import torch.distributed as dist

                box = [uid]
                dist.broadcast_object_list(box, src=0)
                uid = box[0]
            else:
                uid = exchange_id(uid)
        with self.torch.cuda.stream(self.stream):
            self.engine.comm_init(uid, rank, world)
        self.native_comm = True

    # -- storage: this rank's rows (vectorbase.py:115-148, 268-287 over a row-sharded corpus) ------------------------------------------
    def set_rows(self, rows: np.ndarray, row_offset: int, dtype: str = "fp32") -> None:
        """Replace this rank's shard by host rows (float32 [n, dim]); uploaded through the pinned ring (tavb_upload_rows)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        self._dtype_code = _native.TAVB_F16 if dtype == "fp16" else _native.TAVB_F32
        self._dtype_name = "fp16" if dtype == "fp16" else "fp32"
        with self.torch.cuda.stream(self.stream):
            self.engine.ordinal_base = int(row_offset)
            if rows.ndim == 2 and rows.shape[1] > 0:
                self.engine.upload_rows(rows, 0, self._dtype_code, capacity_hint=2 * rows.shape[0])
            else:
                self.engine.clear()

    def append_rows(self, rows: np.ndarray) -> None:
        """Append host rows (float32 [n, dim]) to this rank's shard: only the new rows travel (capacity doubles; a shard adopted from the
        caller's tensor is copied into a buffer of the backend's own first)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        code = self.engine.dtype if self.engine.corpus is not None else getattr(self, "_dtype_code", _native.TAVB_F32)
        with self.torch.cuda.stream(self.stream):
            self.engine.upload_rows(rows, self.engine.rows if self.engine.corpus is not None else 0, code)

    def rows_to_host(self) -> np.ndarray:
        """This rank's rows as float32 [n, dim] (fp16 storage widened)."""
        self.stream.synchronize()
        if self.engine.corpus is None:
            return np.zeros((0, 0), dtype=np.float32)
        return self.engine.corpus[: self.engine.rows].float().cpu().numpy()

    # -- the forms beside the plain lookup: this shard's part, as keys that already carry GLOBAL positions / ordinals ----------------
    def local_search_subset(self, query: np.ndarray, local_rows: np.ndarray, positions: np.ndarray, k: int, thr: float):
        """rows of THIS shard (local numbering) that the caller's subset names, `positions[i]` = where local_rows[i] sits in the
        caller's list -> keys [1, k] carrying those positions (subset gather kernel + position remap, on the backend's stream)."""
        torch = self.torch
        dev = torch.device("cuda", self.device)
        with torch.cuda.stream(self.stream):
            if len(local_rows) == 0:
                return torch.zeros((1, k), dtype=torch.int64, device=dev)
            # (the copies run on the backend's stream, in front of the kernels that read them)
            dq = torch.from_numpy(np.ascontiguousarray(query, dtype=np.float32)).to(dev)
            rows = torch.from_numpy(np.ascontiguousarray(local_rows, dtype=np.int32)).to(dev)
            pmap = torch.from_numpy(np.ascontiguousarray(positions, dtype=np.int32)).to(dev)
            keys = self.engine.search_subset_device(dq, rows, k, thr)
            self.engine.remap_key_positions(keys, pmap)
            self._keep = (dq, rows, pmap)  # alive until the next call (the kernels run asynchronously)
            return keys

    def subset_to_device(self, local_rows: np.ndarray, positions: np.ndarray):
        """This shard's part of a caller's subset as device tensors (rows int32 [S], positions int32 [S]) -- a handle for
        `local_search_subset_resident`; None when no row of the subset lies in this shard."""
        torch = self.torch
        if len(local_rows) == 0:
            return None
        dev = torch.device("cuda", self.device)
        with torch.cuda.stream(self.stream):
            return (torch.from_numpy(np.ascontiguousarray(local_rows, dtype=np.int32)).to(dev),
                    torch.from_numpy(np.ascontiguousarray(positions, dtype=np.int32)).to(dev))

    def local_search_subset_resident(self, query: np.ndarray, handle, k: int, thr: float):
        """`local_search_subset` over a handle of `subset_to_device`: only the query travels."""
        torch = self.torch
        dev = torch.device("cuda", self.device)
        with torch.cuda.stream(self.stream):
            if handle is None:
                return torch.zeros((1, k), dtype=torch.int64, device=dev)
            dq = torch.from_numpy(np.ascontiguousarray(query, dtype=np.float32)).to(dev)
            keys = self.engine.search_subset_device(dq, handle[0], k, thr)
            self.engine.remap_key_positions(keys, handle[1])
            self._keep = (dq, handle)  # alive until the next call (the kernels run asynchronously)
            return keys

    def local_survivors(self, query: np.ndarray, thr: float):
        """every row of this shard with score >= thr -> (global ordinals int64, scores float32), unsorted (one emit-all pass)."""
        return self.engine.search_all(np.ascontiguousarray(query, dtype=np.float32), np.float32(thr), None)

    def keys_to_device(self, keys: np.ndarray):
        torch = self.torch
        with torch.cuda.stream(self.stream):
            return torch.from_numpy(np.ascontiguousarray(keys).view(np.int64)).to(torch.device("cuda", self.device))

    def search_allgather(self, queries, k: int, thr: float):
        """scan -> ncclAllGather -> merge inside libtavb, merged keys written into a reused pinned host buffer: -> int64 [nq, k] (host view,
        valid until the next call)."""
        shape = (int(queries.shape[0]), int(k))
        pinned = self._pinned.get(shape)
        if pinned is None:
            pinned = self._pinned[shape] = self.torch.empty(shape, dtype=self.torch.int64).pin_memory()
        self.engine.search_allgather(queries, k, thr, out_keys=pinned)
        self.engine.synchronize()
        return pinned.numpy()

    def local_search(self, queries, k: int, thr: float):
        with self.torch.cuda.stream(self.stream):
            return self.engine.search_device(queries, k, thr)

    def merge(self, gathered):
        with self.torch.cuda.stream(self.stream):
            return self.engine.merge_device(gathered)

    def to_host(self, keys) -> np.ndarray:
        # pinned staging buffer, reused: one async copy + one stream sync, no allocation per lookup
        shape = tuple(keys.shape)
        pinned = self._pinned.get(shape)
        if pinned is None:
            pinned = self.torch.empty(shape, dtype=self.torch.int64).pin_memory()
            self._pinned[shape] = pinned
        with self.torch.cuda.stream(self.stream):
            pinned.copy_(keys, non_blocking=True)
        self.stream.synchronize()
        return pinned.numpy().copy()

    def failed_lists(self, nq: int, k: int):
        """[nq, k] lists of PEER_FAILED_KEY: what a rank whose local search failed sends into the exchange.  A buffer of their own (never the
        gather buffer: with one rank that IS the collective's output), filled on the backend's stream -- the stream the exchange and the merge
        run on, so the fill is ordered in front of them."""
        buf = self._failed.get((nq, k))
        with self.torch.cuda.stream(self.stream):
            if buf is None:
                buf = self._failed[(nq, k)] = self.torch.empty((nq, k), dtype=self.torch.int64, device=self.torch.device("cuda", self.device))
            buf.fill_(PEER_FAILED_KEY)
        return buf

    def empty_gather(self, world: int, nq: int, k: int):
        shape = (world, nq, k)
        buf = self._gather.get(shape)
        if buf is None:
            with self.torch.cuda.stream(self.stream):
                buf = self.torch.empty(shape, dtype=self.torch.int64, device=self.torch.device("cuda", self.device))
            self._gather[shape] = buf
        return buf


@dataclass
class ShardedResult:
    ordinals: np.ndarray  # int64 [nq, k] global row ordinals
    scores: np.ndarray  # float32 [nq, k]
    counts: np.ndarray  # int32 [nq]


class ShardedSearcher:
    """Collective top-k over a row-sharded corpus.  Every rank calls `search` with the
    same queries (they are tiny next to the corpus: <= 3 MiB for 1024 x 1536 fp16, so the
    caller broadcasts/duplicates them) and every rank gets the same global result."""

    def __init__(self, backend: ShardBackend, group=None, always_collective: bool = False, gather_fn=None):
        import torch.distributed as dist

        self.dist = dist
        self.backend = backend
        self.group = group
        self.always_collective = always_collective  # run the all-gather even for one rank (tests)
        self.gather_fn = gather_fn  # test hook: (local [nq,k]) -> gathered [world,nq,k] by other means than RCCL
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def search_keys(self, queries, k: int, min_score: float = 0.0):
        """-> backend tensor int64 [nq, k] of merged keys (still on the device, async)."""
        if not (1 <= k <= _native.MAX_FUSED_K):
            raise ValueError(f"k must be in 1..{_native.MAX_FUSED_K}")
        thr = float(_native.f32_threshold(min_score))
        collective = not (self.world == 1 and not (self.always_collective and self.dist.is_initialized()))
        failure = None
        try:
            local = self.backend.local_search(queries, k, thr)
        except Exception as exc:  # noqa: BLE001 -- whatever the backend raised: the peers must not be left waiting in the all-gather
            if not collective:
                raise
            # the protocol of tavb_search_allgather (include/tavb.h): this rank joins the exchange with TAVB_KEY_PEER_FAILED in every slot -- it
            # sorts above every real key, so it leads every merged list on every rank -- and raises its own error afterwards
            failure = exc
            local = self.backend.failed_lists(int(queries.shape[0]), k)
        if not collective:
            return local
        nq = local.shape[0]
        if self.gather_fn is not None:
            merged = self.backend.merge(self.gather_fn(local))
        else:
            gathered = self.backend.empty_gather(self.world, nq, k)
            stream = getattr(self.backend, "stream", None)
            if stream is not None:
                with self.backend.torch.cuda.stream(stream):
                    self.dist.all_gather_into_tensor(gathered.view(-1), local.contiguous().view(-1), group=self.group)
            else:
                self.dist.all_gather_into_tensor(gathered.view(-1), local.contiguous().view(-1), group=self.group)
            merged = self.backend.merge(gathered)
        if failure is not None:
            raise failure
        return merged

    def exchange(self, local_keys):
        """this rank's sorted lists [nq, k] (global ordinals / positions) -> the lists merged over all ranks, on every rank."""
        if getattr(self.backend, "native_comm", False) and self.gather_fn is None:
            return self.backend.engine.allgather_merge(local_keys)
        if self.gather_fn is not None:
            return self.backend.merge(self.gather_fn(local_keys))
        if self.world == 1 and not (self.always_collective and self.dist.is_initialized()):
            return local_keys
        nq, k = local_keys.shape
        gathered = self.backend.empty_gather(self.world, nq, k)
        self.dist.all_gather_into_tensor(gathered.view(-1), local_keys.contiguous().view(-1), group=self.group)
        return self.backend.merge(gathered)

    def search(self, queries, k: int, min_score: float = 0.0) -> ShardedResult:
        if getattr(self.backend, "native_comm", False) and self.gather_fn is None:
            # the product path: one C-ABI call (no torch.distributed, no torch op), results land in pinned host memory
            if not (1 <= k <= _native.MAX_FUSED_K):
                raise ValueError(f"k must be in 1..{_native.MAX_FUSED_K}")
            keys = self.backend.search_allgather(queries, k, float(_native.f32_threshold(min_score)))
        else:
            keys = self.backend.to_host(self.search_keys(queries, k, min_score))
        if keys.size and (np.asarray(keys).reshape(keys.shape[0], -1)[:, 0].view(np.int64) == PEER_FAILED_KEY).any():
            raise PeerFailedError("a rank of the collective lookup failed in its local search: the merged lists are missing its shard")
        ords, scs, cnts = _native.decode_keys(keys)
        return ShardedResult(ords, scs, cnts)


class ShardedVectorBase:
    """VectorBase-shaped front end over a row-sharded corpus: every rank holds rows
    [row_offset, row_offset + local_rows) and every lookup is a collective call (all ranks pass the same
    query, all ranks get the same global answer).  Covers the lookup methods of the reference class
    (vectorbase.py:163-230: plain, batched, predicate and subset forms) and its storage methods (`add_embedding(s)`: appends go to the
    last rank's shard; `serialize` / `deserialize`: per-rank matrices; `clear`) -- all collective: every rank makes the same call."""

    def __init__(self, backend: ShardBackend, row_offset: int, local_rows: int, total_rows: int, group=None):
        self.backend = backend
        self.row_offset = int(row_offset)
        self.local_rows = int(local_rows)
        self.total_rows = int(total_rows)
        self.searcher = ShardedSearcher(backend, group=group)
        self._storage_dtype = backend.storage_dtype() if hasattr(backend, "storage_dtype") else "fp32"
        self._subset_cache = None  # (caller's subset object, private copy, total rows, ordinals int64, the backend's handle) of the last subset lookup

    def __len__(self) -> int:
        return self.total_rows

    def __bool__(self) -> bool:
        return True

    # ---- storage (collective: every rank makes the same call with the same arguments) ------------------------------------------------
    @property
    def _world(self) -> int:
        return self.searcher.world

    @property
    def _rank(self) -> int:
        return self.searcher.rank

    def add_embeddings(self, keys, embeddings) -> None:
        """vectorbase.py:130-148 over row shards: the new rows get the next global ordinals and go to the LAST rank's shard (contiguous
        ranges stay contiguous; only that rank uploads anything, and only the new rows); every rank advances its row count.  `keys` is
        accepted for signature compatibility (the reference caches key -> embedding in its model; there is no model here)."""
        rows = np.asarray(embeddings, dtype=np.float32)
        if rows.ndim != 2:
            raise ValueError(f"Expected 2D embeddings array, got {rows.ndim}D")
        if keys is not None and len(keys) != len(rows):
            raise ValueError(f"Number of keys {len(keys)} does not match number of embeddings {len(rows)}")
        if len(rows) == 0:
            return
        if self._rank == self._world - 1:
            self.backend.append_rows(rows)
            self.local_rows += len(rows)
        self.total_rows += len(rows)
        self._subset_cache = None
        # Appends always land on the last rank: contiguous ranges stay contiguous and nothing already placed moves -- but every collective
        # lookup waits for the largest shard, so an index GROWN by appends drifts towards single-GPU speed.  `imbalance()` says how far it has
        # drifted; `rebalance()` re-deals the rows (a collective: every rank hands its rows to `deserialize` of the balanced layout).

    def imbalance(self) -> float:
        """Collective: largest shard / mean shard (1.0 = balanced)."""
        if self.total_rows == 0 or self._world == 1:
            return 1.0
        counts = [None] * self._world
        self.searcher.dist.all_gather_object(counts, self.local_rows, group=self.searcher.group)
        return max(counts) / (self.total_rows / self._world)

    def rebalance(self, dtype: str | None = None) -> None:
        """Collective: re-deal the rows into balanced contiguous shards (`shard_range`), stored as they are now (`dtype` None: the dtype handed
        to `deserialize` / of the adopted shard tensor -- an fp16 index stays fp16: twice the HBM and other kernels otherwise).  Every rank serialises its rows, the ranks exchange
        them over torch.distributed (host side, every rank sees the whole matrix for a moment: a maintenance step for corpora that fit host memory,
        not the lookup path -- bigger ones are re-sharded from their source with `deserialize`) and each keeps its new range."""
        mine = self.serialize()
        if dtype is None:
            dtype = self._storage_dtype
        if self._world == 1:
            return
        parts = [None] * self._world
        self.searcher.dist.all_gather_object(parts, (self.row_offset, mine), group=self.searcher.group)
        parts = [p for p in parts if p[1].size]
        dim = parts[0][1].shape[1] if parts else 0
        whole = np.concatenate([p[1] for p in sorted(parts, key=lambda t: t[0])]) if parts else np.zeros((0, dim), np.float32)
        lo, hi = shard_range(len(whole), self._world, self._rank)
        self.deserialize(whole[lo:hi], dtype)

    def add_embedding(self, key, embedding) -> None:
        """vectorbase.py:115-128."""
        row = np.asarray(embedding, dtype=np.float32)
        if row.ndim == 1:
            row = row[None, :]
        if row.ndim != 2 or row.shape[0] != 1:
            raise ValueError(f"Expected a single embedding, got shape {row.shape}")
        self.add_embeddings(None if key is None else [key], row)

    def serialize(self) -> np.ndarray:
        """vectorbase.py:268-271, per rank: THIS rank's rows [row_offset, row_offset + local_rows) as a float32 matrix (a copy: the rows
        live on the device); `row_offset`, `local_rows`, `total_rows` place it in the whole.  Concatenating the ranks' matrices in rank
        order is the reference's `serialize()` of the whole index."""
        return self.backend.rows_to_host()

    def deserialize(self, local_data, dtype: str = "fp32") -> None:
        """vectorbase.py:273-287, per rank: every rank hands in ITS rows (float32 [n_r, dim], None or empty for none); the ranks agree on
        the offsets (an all-gather of the row counts over torch.distributed) and each uploads its own shard."""
        rows = np.zeros((0, 0), dtype=np.float32) if local_data is None else np.asarray(local_data, dtype=np.float32)
        if rows.ndim != 2:
            raise ValueError(f"Expected 2D embeddings array, got {rows.ndim}D")
        counts = [len(rows)]
        if self._world > 1:
            counts = [None] * self._world
            self.searcher.dist.all_gather_object(counts, len(rows), group=self.searcher.group)
        self.row_offset = int(sum(counts[: self._rank]))
        self.local_rows = len(rows)
        self.total_rows = int(sum(counts))
        self._storage_dtype = "fp16" if dtype == "fp16" else "fp32"
        self._subset_cache = None
        self.backend.set_rows(rows, self.row_offset, dtype)  # (no rows: the backend drops its shard)

    def clear(self) -> None:
        """vectorbase.py:248-255."""
        self.deserialize(None)

    @classmethod
    def from_device_shard(cls, device: int, shard_tensor, row_offset: int, total_rows: int, group=None):
        import torch.distributed as dist

        backend = DeviceShardBackend(device)
        backend.set_shard(shard_tensor, row_offset=row_offset)
        if dist.is_initialized() and group is None:
            backend.init_comm(dist.get_rank(), dist.get_world_size())
        return cls(backend, row_offset, shard_tensor.shape[0], total_rows, group=group)

    def _queries(self, q):
        a = np.ascontiguousarray(q, dtype=np.float32)
        if hasattr(self.backend, "stage_queries"):
            return self.backend.stage_queries(a)
        torch = getattr(self.backend, "torch", None)
        if torch is None:
            import torch as _t

            return _t.from_numpy(a)
        return torch.from_numpy(a).to(torch.device("cuda", self.backend.device))

    def fuzzy_lookup_embeddings(self, embeddings, max_hits: int | None = None, min_score: float | None = None):
        if max_hits is None:
            max_hits = 10
        if min_score is None:
            min_score = 0.0
        q = np.asarray(embeddings, dtype=np.float32)
        if q.ndim != 2:
            raise ValueError(f"Expected 2D embeddings array, got {q.ndim}D")
        if self.total_rows == 0 or len(q) == 0:
            return [[] for _ in range(len(q))]
        res = self.searcher.search(self._queries(q), max_hits, min_score)
        from .vectorbase import _scored_lists  # (the lists are built in C: csrc/tavb_pyhits.c)

        return _scored_lists(res.ordinals, res.scores, res.counts, int(max_hits))

    def fuzzy_lookup_embedding(self, embedding, max_hits: int | None = None, min_score: float | None = None, predicate=None):
        """vectorbase.py:163-201.  With a predicate (:191-201): every rank applies it to the survivors of ITS shard, in ascending
        ordinal order (so the predicate is called once per survivor across the job, not once per survivor per rank), keeps its best
        `max_hits`, and the per-rank lists are merged like any other -- every rank returns the whole-corpus answer."""
        if predicate is None:
            return self.fuzzy_lookup_embeddings(np.asarray(embedding, dtype=np.float32)[None, :], max_hits, min_score)[0]
        from .vectorbase import ScoredInt

        k = 10 if max_hits is None else int(max_hits)
        if not (1 <= k <= _native.MAX_FUSED_K):
            raise ValueError(f"max_hits must be in 1..{_native.MAX_FUSED_K} for a sharded lookup with a predicate")
        thr = _native.f32_threshold(0.0 if min_score is None else min_score)
        if self.total_rows == 0:
            return []
        q = np.ascontiguousarray(embedding, dtype=np.float32)

        def local_lists():
            ids, scs = (np.zeros(0, np.int64), np.zeros(0, np.float32)) if self.local_rows == 0 else self.backend.local_survivors(q, thr)
            order = np.lexsort((ids,))  # ascending ordinal: the order of np.flatnonzero (:193)
            keep = np.fromiter((bool(predicate(int(i))) for i in ids[order]), dtype=bool, count=len(order))
            ids, scs = ids[order][keep], scs[order][keep]
            keys = (scs.astype(np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - ids.astype(np.uint64))
            local = np.zeros((1, k), dtype=np.uint64)
            best = np.sort(keys)[::-1][:k]
            local[0, : len(best)] = best
            return self.backend.keys_to_device(local)

        ords, sc, cnt = self._exchange_or_fail(local_lists, 1, k)
        return [ScoredInt(int(o), float(s_)) for o, s_ in zip(ords[0, : cnt[0]].tolist(), sc[0, : cnt[0]].tolist())]

    def _exchange_or_fail(self, local_lists, nq: int, k: int):
        """The collective forms beside the plain lookup (subset, predicate): `local_lists()` -> this rank's [nq, k] keys -> exchange -> decoded
        merged lists.  The failure protocol of `ShardedSearcher.search_keys` / tavb_search_allgather: a rank whose local part raises -- the
        caller's predicate included -- still joins the exchange, with PEER_FAILED_KEY lists, then raises ITS error; the others raise
        `PeerFailedError` instead of waiting for ever or returning an answer that misses a shard."""
        failure = None
        try:
            local = local_lists()
        except Exception as exc:  # noqa: BLE001
            if self._world == 1 and not getattr(self.backend, "native_comm", False):
                raise
            failure = exc
            local = self.backend.failed_lists(nq, k)
        merged = self.backend.to_host(self.searcher.exchange(local))
        if failure is not None:
            raise failure
        if merged.size and (np.asarray(merged).reshape(merged.shape[0], -1)[:, 0].view(np.int64) == PEER_FAILED_KEY).any():
            raise PeerFailedError("a rank of the collective lookup failed in its local part: the merged lists are missing its shard")
        return _native.decode_keys(merged)

    def fuzzy_lookup_embedding_in_subset(self, embedding, ordinals_of_subset, max_hits: int | None = None, min_score: float | None = None):
        """vectorbase.py:203-230 over the row-sharded corpus: every rank gathers the rows of the caller's subset that lie in its shard,
        the per-rank top-k lists (keys carrying POSITIONS in the caller's list: duplicates and negative, wrapping ordinals behave as in
        the reference) are merged over the ranks; returns the caller's ordinals."""
        from .vectorbase import ScoredInt

        k = 10 if max_hits is None else int(max_hits)
        if not (1 <= k <= _native.MAX_FUSED_K):
            raise ValueError(f"max_hits must be in 1..{_native.MAX_FUSED_K} for a sharded subset lookup")
        thr = float(_native.f32_threshold(0.0 if min_score is None else min_score))
        if len(ordinals_of_subset) == 0 or self.total_rows == 0:
            return []
        q = np.ascontiguousarray(embedding, dtype=np.float32)
        # the same subset object with the same content again (the memory provider's scope list per query term,
        # storage/memory/messageindex.py:173-183): this shard's part of it stays on the device (as VectorBase does it)
        layout = (self.total_rows, self.row_offset, self.local_rows)
        cached = self._subset_cache
        resident = hasattr(self.backend, "subset_to_device") and isinstance(ordinals_of_subset, (list, np.ndarray))
        if (resident and cached is not None and cached[0] is ordinals_of_subset and cached[2] == layout and len(cached[1]) == len(ordinals_of_subset)
                and (np.array_equal(cached[1], ordinals_of_subset) if isinstance(ordinals_of_subset, np.ndarray) else cached[1] == ordinals_of_subset)):
            subset = cached[3]
            handle = cached[4]

            def local_lists():
                return self.backend.local_search_subset_resident(q, handle, k, thr)
        else:
            subset = np.asarray(ordinals_of_subset)
            if subset.dtype.kind not in "iu":
                raise IndexError("arrays used as indices must be of integer (or boolean) type")
            subset = subset.astype(np.int64, copy=False).reshape(-1)
            n = self.total_rows
            rows = np.where(subset < 0, subset + n, subset)  # numpy index wrap (:218)
            bad = (rows < 0) | (rows >= n)
            if bad.any():
                raise IndexError(f"index {int(subset[np.argmax(bad)])} is out of bounds for axis 0 with size {n}")
            mine = np.flatnonzero((rows >= self.row_offset) & (rows < self.row_offset + self.local_rows))  # ascending positions: lists stay sorted among ties
            # (argument errors above are the same on every rank -- every rank holds the same list -- and raise before anything collective)
            if resident:
                def local_lists():
                    handle = self.backend.subset_to_device(rows[mine] - self.row_offset, mine)
                    is_array = isinstance(ordinals_of_subset, np.ndarray)
                    keep = subset.copy() if is_array else subset
                    self._subset_cache = (ordinals_of_subset, ordinals_of_subset.copy() if is_array else list(ordinals_of_subset), layout, keep, handle)
                    return self.backend.local_search_subset_resident(q, handle, k, thr)
            else:
                def local_lists():
                    return self.backend.local_search_subset(q, rows[mine] - self.row_offset, mine, k, thr)
        pos, sc, cnt = self._exchange_or_fail(local_lists, 1, k)
        return [ScoredInt(int(subset[p]), float(s_)) for p, s_ in zip(pos[0, : cnt[0]].tolist(), sc[0, : cnt[0]].tolist())]

    def lookup_messages_by_embedding(self, embedding, row_to_message, max_matches: int | None = None, threshold_score: float | None = None, accept=None):
        """`SqliteMessageTextIndex.lookup_by_embedding` / `lookup_in_subset_by_embedding` (storage/sqlite/messageindex.py:296-326, 182-257)
        over the sharded corpus: the whole-corpus top-`max_matches` chunk rows (collective), THEN the provider's message filter and
        best-score-per-message aggregation on the merged hits (`row_to_message`: the GLOBAL chunk row -> message map; identical on
        every rank, so every rank returns the same messages)."""
        from .adapters import best_score_per_message

        hits = self.fuzzy_lookup_embedding(embedding, max_hits=max_matches, min_score=threshold_score)
        members = None if accept is None else (accept if callable(accept) else set(int(x) for x in accept).__contains__)
        return best_score_per_message(hits, row_to_message, max_matches, members)
