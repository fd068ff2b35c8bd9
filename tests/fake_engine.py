"""TEST INFRASTRUCTURE -- numpy stand-in for `typeagent_py_amd._native.Engine`, so that the HOST side of the drop-in class
(bookkeeping, paging, sharding, the reference's own consumers running on top of it) can be exercised in a container
without a GPU.  It answers a lookup with the oracle's arithmetic and packs / orders results exactly like the kernels do
(score descending, ordinal ascending).  The product never uses it: tests monkeypatch `_native.Engine` explicitly."""

from __future__ import annotations

import numpy as np

from oracle import vectorbase_oracle as vo
from typeagent_py_amd import _native


def _order(scores: np.ndarray, ids: np.ndarray, k: int):
    order = np.lexsort((ids, -scores.astype(np.float64)))[:k]
    return ids[order].astype(np.int64), scores[order].astype(np.float32)


class FakeEngine:
    instances: list = []

    def __init__(self, device=None, use_torch_stream=False):
        self.device = device
        self.corpus = None
        self.rows = 0
        self.dim = 0
        self.dtype = _native.TAVB_F32
        self.ordinal_base = 0
        self._pending = None
        self.row_messages = None
        self.uploads: list = []  # (start, count) per upload_rows call
        self.options: dict = {}
        FakeEngine.instances.append(self)

    def close(self):
        pass

    def clear(self):
        self.rows = 0

    def set_option(self, name, value):
        self.options[name] = value

    def get_option(self, name):
        return self.options.get(name, 0)

    def upload_rows(self, host_rows, start, dtype, capacity_hint=0):
        n_new = start + host_rows.shape[0]
        if self.corpus is None or self.corpus.shape[0] < n_new or self.corpus.shape[1] != host_rows.shape[1]:
            fresh = np.zeros((max(n_new, capacity_hint, 4), host_rows.shape[1]), dtype=np.float32)
            if start:
                fresh[:start] = self.corpus[:start]
            self.corpus = fresh
        rows = np.asarray(host_rows, dtype=np.float32)
        self.corpus[start:n_new] = rows.astype(np.float16).astype(np.float32) if dtype == _native.TAVB_F16 else rows
        self.rows, self.dim, self.dtype = n_new, host_rows.shape[1], dtype
        self.uploads.append((start, host_rows.shape[0]))
        return True

    def _scores(self, q):
        return vo.scores_full(self.corpus[: self.rows], np.asarray(q, dtype=np.float32))

    # -- synchronous forms -----------------------------------------------------------------------------------------
    def search(self, q, k, thr, after=None):
        sc = self._scores(q)
        ids = np.arange(self.rows, dtype=np.int64) + self.ordinal_base
        ok = sc >= thr
        if after is not None:
            ok &= (sc < np.float32(after[0])) | ((sc == np.float32(after[0])) & (ids > after[1]))
        return _order(sc[ok], ids[ok], k)

    def search_all(self, q, thr, max_out=None, subset_rows=None):
        if subset_rows is not None:
            pos, sc = self.search_subset(q, subset_rows, len(subset_rows) if max_out is None else max_out, thr)
            return pos, sc
        return self.search(q, self.rows if max_out is None else max_out, thr)

    def search_batch(self, queries, k, thrs):
        queries = np.asarray(queries, dtype=np.float32)
        t = np.broadcast_to(np.asarray(thrs, dtype=np.float32), (len(queries),))
        ords = np.zeros((len(queries), k), dtype=np.int64)
        scs = np.zeros((len(queries), k), dtype=np.float32)
        cnts = np.zeros(len(queries), dtype=np.int32)
        for i, q in enumerate(queries):
            o, s = self.search(q, k, t[i])
            ords[i, : len(o)], scs[i, : len(o)], cnts[i] = o, s, len(o)
        return ords, scs, cnts

    def search_subset(self, q, rows, k, thr, after=None):
        rows = np.asarray(rows, dtype=np.int64)
        sc = vo.cosine_to_score(np.dot(self.corpus[rows], np.asarray(q, dtype=np.float32)))
        pos = np.arange(len(rows), dtype=np.int64)
        ok = sc >= thr
        if after is not None:
            ok &= (sc < np.float32(after[0])) | ((sc == np.float32(after[0])) & (pos > after[1]))
        return _order(sc[ok], pos[ok], k)

    def rows_to_device(self, rows):  # the "device" copy of a subset's row list: a private int32 array, counted
        self.subset_uploads = getattr(self, "subset_uploads", 0) + 1
        return np.array(rows, dtype=np.int32)

    def search_subset_resident(self, q, dev_rows, k, thr):
        return self.search_subset(q, dev_rows.astype(np.int64), k, thr)

    # -- split form (device groups) ----------------------------------------------------------------------------------
    def _keys(self, q, k, thr, bound=None):
        sc = self._scores(q)
        keys = (sc.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - (np.arange(self.rows, dtype=np.uint64) + np.uint64(self.ordinal_base)))
        ok = sc >= thr
        if bound is not None:
            ok &= keys < np.uint64(bound)
        keys = np.sort(keys[ok])[::-1][:k]
        out = np.zeros(k, dtype=np.uint64)
        out[: len(keys)] = keys
        return out

    def search_begin(self, queries, k, thrs, cursor_key=None):
        self._pending = np.stack([self._keys(q, k, t, cursor_key) for q, t in zip(queries, thrs)])

    def search_end(self, nq, k, out_keys):
        assert self._pending.shape == (nq, k)
        out_keys[...] = self._pending
        self._pending = None

    # -- message re-rank -------------------------------------------------------------------------------------------
    def set_row_messages(self, row_to_message):
        self.row_messages = np.asarray(row_to_message, dtype=np.int64)

    def search_messages(self, q, k, thr, max_messages, accept=None, subset_rows=None):
        from oracle import messages_oracle as mo

        if subset_rows is not None:
            pos, sc = self.search_subset(q, subset_rows, k, thr)
            hits = [(int(np.asarray(subset_rows)[p]), float(s)) for p, s in zip(pos, sc)]
        else:
            o, sc = self.search(q, k, thr)
            hits = [(int(i), float(s)) for i, s in zip(o, sc)]
        members = None if accept is None else set(int(x) for x in accept)
        out = mo.sqlite_messages_from_hits(hits, self.row_messages, None if members is None else members.__contains__, max_messages)
        return np.asarray([m for m, _ in out], dtype=np.int64), np.asarray([s for _, s in out], dtype=np.float32)
