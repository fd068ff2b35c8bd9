"""One VectorBase over several GPUs of a node, driven from ONE process and ONE thread.

typeagent is a single asyncio process whose query pipeline calls `fuzzy_lookup_embedding*` synchronously
(knowpro/query.py:886-934 of the reference), so the row-sharded corpus has to sit under one Python object: the SPMD
form (`sharded.py`, one process per GPU + RCCL) needs every rank to call collectively and is for the bench / servers.

`DeviceGroup` looks like `_native.Engine` to `VectorBase` (same methods), and holds one libtavb context + stream per
device.  Rows are sharded contiguously: shard g = rows [g*B, (g+1)*B), the last shard takes the appends (rebalanced
when it reaches twice the block size).  A lookup copies the query to every device and enqueues the per-shard scans back
to back (`tavb_search_begin`: nothing blocks), then collects the per-shard key lists (`tavb_search_end`), merges them
on the host (`tavb_merge_keys_host`: G x k keys per query -- cheaper than any collective) and
decodes.  Keys carry global ordinals and order by (score desc, ordinal asc), so the merged answer is the whole-corpus
answer, ties included.
"""

from __future__ import annotations

import functools
import threading

import numpy as np

from . import _native


def _locked(method):
    """Hold the group's lock for the whole call: a lookup is `search_begin` on every device followed by `search_end` on every device, and
    each engine only locks per call -- two threads interleaving those pairs would collect each other's results (or trip the
    'tavb_search_end does not match the pending tavb_search_begin' check)."""

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with self._lock:
            return method(self, *args, **kwargs)

    return wrapper


class DeviceGroup:
    def __init__(self, devices: list[int]):
        if len(devices) < 1:
            raise ValueError("devices must name at least one GPU")
        self.devices = [int(d) for d in devices]
        self.engines = [_native.Engine(d) for d in self.devices]
        self._lock = threading.RLock()
        self.block = 0  # rows per shard (the last shard may hold more)
        self.bounds = [0] * (len(self.devices) + 1)  # shard g = rows [bounds[g], bounds[g+1])
        self.rows = 0
        self.dim = 0
        self.dtype = _native.TAVB_F32
        self.ordinal_base = 0
        self.corpus = None  # truthy once rows are resident (VectorBase only tests it against None)

    # -- lifecycle / options -------------------------------------------------------------------------------------
    def close(self) -> None:
        for e in self.engines:
            e.close()

    def set_option(self, name: str, value: int) -> None:
        for e in self.engines:
            e.set_option(name, value)

    def get_option(self, name: str) -> int:
        vals = [e.get_option(name) for e, lo, hi in self._active()] or [self.engines[0].get_option(name)]
        return max(vals) if name in ("last_flagged",) else vals[0]

    def profile_enable(self, on: bool = True) -> None:
        for e in self.engines:
            e.profile_enable(on)

    def profile_reset(self) -> None:
        for e in self.engines:
            e.profile_reset()

    def profile_read(self, kernel_id: int) -> tuple[float, int]:
        """(max over devices of the summed kernel time, launches on the first active device)"""
        parts = [e.profile_read(kernel_id) for e, lo, hi in self._active()] or [(0.0, 0)]
        return max(p[0] for p in parts), parts[0][1]

    def synchronize(self) -> None:
        for e in self.engines:
            e.synchronize()

    def _active(self):
        return [(e, self.bounds[g], self.bounds[g + 1]) for g, e in enumerate(self.engines) if self.bounds[g + 1] > self.bounds[g]]

    # -- corpus --------------------------------------------------------------------------------------------------
    def _layout(self, n: int, block: int) -> list[int]:
        g = len(self.engines)
        b = [min(i * block, n) for i in range(g)] + [n]
        return b

    @_locked
    def upload_rows(self, host_rows: np.ndarray, start: int, dtype: int, capacity_hint: int = 0) -> bool:
        """Make rows [start, start + len) of the sharded device copy equal `host_rows`.  Returns False (nothing done) when
        the shard layout has to change and the caller must upload from row 0 instead."""
        g = len(self.engines)
        n_new = start + host_rows.shape[0]
        dim = host_rows.shape[1]
        fresh = start == 0 or self.corpus is None or dim != self.dim or dtype != self.dtype
        if fresh and start != 0:
            return False
        if fresh:
            # balanced: ceil(n / g) rows per shard, every device reserving twice that; appends go to the last shard until it outgrows its
            # reservation, then the layout is rebuilt balanced.  (Round 3 doubled the BLOCK at a re-shard to re-shard less often: right after
            # one, 5 of 8 devices held rows and lookups ran ~1.8x slower until the index had doubled -- round-3 advice.  Balanced layouts
            # re-shard every time the index grows by 1/g: O(g) uploads per appended row when an index is grown row by row, 1 when it is
            # built in bulk; lookups always run on g equal shards.)
            self.block = max(1, -(-max(n_new, 1) // g))
        elif n_new > (g + 1) * self.block:  # the last shard would exceed twice the block: rebalance
            return False
        bounds = self._layout(n_new, self.block)
        for gi, e in enumerate(self.engines):
            lo, hi = bounds[gi], bounds[gi + 1]
            if hi <= lo:
                if e.rows:
                    e.clear()
                continue
            a = max(lo, start)
            if a >= hi and not fresh and e.rows == hi - lo:
                continue  # untouched shard
            e.ordinal_base = lo
            local = host_rows[a - start : hi - start] if a < hi else host_rows[:0]
            cap = 2 * self.block
            e.upload_rows(local, a - lo, dtype, capacity_hint=cap if (fresh or e.corpus is None) else 0)
        self.bounds, self.rows, self.dim, self.dtype = bounds, n_new, dim, dtype
        self.corpus = True
        return True

    @_locked
    def set_shard_tensors(self, tensors, ordinal_base: int = 0) -> None:
        """Adopt one device tensor per GPU (row shards in order) without a host copy."""
        if len(tensors) != len(self.engines):
            raise ValueError("one tensor per device")
        lo = 0
        bounds = [0]
        for e, t in zip(self.engines, tensors):
            e.set_corpus_tensor(t, ordinal_base=ordinal_base + lo)
            lo += int(t.shape[0])
            bounds.append(lo)
        self.bounds, self.rows = bounds, lo
        self.dim, self.dtype = self.engines[0].dim, self.engines[0].dtype
        self.block = max(1, max(b - a for a, b in zip(bounds, bounds[1:])))
        self.ordinal_base = ordinal_base
        self.corpus = list(tensors)

    @_locked
    def clear(self) -> None:
        for e in self.engines:
            e.clear()
        self.rows = 0
        self.bounds = [0] * (len(self.engines) + 1)

    # -- lookups -------------------------------------------------------------------------------------------------
    def _query(self, q) -> np.ndarray:
        a = np.ascontiguousarray(q, dtype=np.float32)
        if a.ndim != 1 or a.shape[0] != self.dim:
            raise ValueError(f"shapes ({self.rows},{self.dim}) and {tuple(np.shape(q))} not aligned: query must have {self.dim} elements")
        return a

    @_locked
    def _gather(self, queries: np.ndarray, k: int, thrs: np.ndarray) -> np.ndarray:
        """-> merged uint64 [nq, k] keys over all shards"""
        act = self._active()
        nq = queries.shape[0]
        if not act:
            return np.zeros((nq, k), dtype=np.uint64)
        for e, lo, hi in act:  # enqueue everywhere first: the devices scan concurrently
            e.search_begin(queries, k, thrs)
        keys = np.empty((len(act), nq, k), dtype=np.uint64)
        for i, (e, lo, hi) in enumerate(act):
            e.search_end(nq, k, keys[i])
        return keys[0] if len(act) == 1 else _native.merge_keys(keys)

    def search(self, q, k: int, thr: np.float32):
        a = self._query(q)[None, :]
        ords, scs, cnts = _native.decode_keys(self._gather(a, k, np.asarray([thr], dtype=np.float32)))
        m = int(cnts[0])
        return ords[0, :m], scs[0, :m]

    def search_batch(self, queries, k: int, thrs):
        a = np.ascontiguousarray(queries, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] != self.dim:
            raise ValueError(f"queries must be [nq, {self.dim}]")
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(thrs, dtype=np.float32), (a.shape[0],)))
        return _native.decode_keys(self._gather(a, k, t))

    @_locked
    def search_all(self, q, thr: np.float32, max_out: int | None = None, subset_rows=None):
        """Every survivor, best first (one emit-all pass per shard, merged on the host)."""
        a = self._query(q)
        ids_all, sc_all = [], []
        rows = None if subset_rows is None else np.ascontiguousarray(subset_rows, dtype=np.int64)
        for e, lo, hi in self._active():
            if rows is None:
                i, s = e.search_all(a, thr, max_out)  # ordinals are global (ordinal_base = the shard's first row)
            else:
                idx = np.flatnonzero((rows >= lo) & (rows < hi))
                if idx.size == 0:
                    continue
                p, s = e.search_all(a, thr, max_out, subset_rows=rows[idx] - lo)
                i = idx[p]
            ids_all.append(i)
            sc_all.append(s)
        if not ids_all:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float32)
        ids, sc = np.concatenate(ids_all), np.concatenate(sc_all)
        order = np.lexsort((ids, -sc.astype(np.float64)))
        if max_out is not None:
            order = order[:max_out]
        return ids[order], sc[order]

    @_locked
    def search_subset(self, q, rows: np.ndarray, k: int, thr: np.float32):
        """rows: int64 global corpus row per subset position -> (positions int64[m], scores float32[m]); the order is
        (score desc, position asc) like one device's."""
        a = self._query(q)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        pos_all, sc_all = [], []
        for e, lo, hi in self._active():
            idx = np.flatnonzero((rows >= lo) & (rows < hi))
            if idx.size == 0:
                continue
            p, s = e.search_subset(a, rows[idx] - lo, k, thr)
            pos_all.append(idx[p])
            sc_all.append(s)
        if not pos_all:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float32)
        pos = np.concatenate(pos_all)
        sc = np.concatenate(sc_all)
        order = np.lexsort((pos, -sc.astype(np.float64)))[:k]
        return pos[order], sc[order]
