"""One device submission for all VectorBase lookups of one user question.

What the reference does per `ConversationBase.query()` (/root/reference):
  * T related-term lookups, sequential, each `fuzzy_lookup(term, max_hits=50, min_score=0.85)` on the
    terms corpus (storage/memory/reltermsindex.py:183-192 -> :320-332; settings convsettings.py:61-63);
  * one message re-rank: `lookup_in_subset_by_embedding` on the message-chunk corpus, a true subset
    gather in the memory provider (knowpro/textlocindex.py:164-177 -> vectorbase.py:203-230) or a full
    scan in the sqlite provider (storage/sqlite/messageindex.py:296-326), k = max_message_matches = 25
    (conversation_base.py:570), threshold 0.7 (convsettings.py:65-66);
  * zero or one thread lookup (storage/memory/convthreads.py:27-44), top-10 at the message-text
    threshold 0.7 (storage/memory/provider.py:64-65).
That is up to T + 2 synchronous numpy passes.  Here the T + 2 query vectors go to the GPU in one
copy, every scan is launched asynchronously on one stream (the T term queries share ONE pass over the
terms corpus), all result keys land in one buffer, and there is one device-to-host copy and one
stream synchronisation.  Results are identical to the separate calls (tests/test_gpu_parity.py).

BASELINE.json config 5 ("fused multi-index query") is defined on top of this class (bench.py
--workload cfg5), because the reference's tools/benchmark_query.py does not touch VectorBase at all
(correction 2).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _native
from .vectorbase import ScoredInt


@dataclass
class FusedResult:
    terms: list[list[ScoredInt]]  # one list per term query
    messages: list[ScoredInt]  # ordinals of the message corpus (subset: the caller's ordinals)
    threads: list[ScoredInt]


class FusedIndexQuery:
    """Holds up to three corpora (device tensors) behind one engine / one stream."""

    TERMS_K, TERMS_MIN = 50, 0.85
    MESSAGES_K, MESSAGES_MIN = 25, 0.7
    THREADS_K, THREADS_MIN = 10, 0.7

    def __init__(self, device: int = 0):
        import torch

        self.torch = torch
        self.device = int(device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self.stream):
            self.engine = _native.Engine(self.device, use_torch_stream=True)
        self.corpora: dict[str, tuple] = {}
        self._pinned_q = None
        self._dev_q = None
        self._pinned_keys = None

    def set_corpus(self, name: str, tensor, rows: int | None = None) -> None:
        """name in {"terms", "messages", "threads"}; tensor: contiguous f32/f16 [N, D] on this device."""
        if name not in ("terms", "messages", "threads"):
            raise ValueError("corpus name must be terms, messages or threads")
        n = tensor.shape[0] if rows is None else int(rows)
        self.corpora[name] = (tensor, n)

    def _use(self, name: str) -> bool:
        entry = self.corpora.get(name)
        if entry is None or entry[1] == 0:
            return False
        self.engine.set_corpus_tensor(entry[0], rows=entry[1], ordinal_base=0, sync_torch=False)
        return True

    def run(self, term_queries, message_query=None, thread_query=None, message_subset=None) -> FusedResult:
        torch = self.torch
        tq = np.ascontiguousarray(term_queries, dtype=np.float32).reshape(-1, np.shape(term_queries)[-1]) if len(term_queries) else np.zeros((0, 0), np.float32)
        T = tq.shape[0]
        dim = tq.shape[1] if T else (len(message_query) if message_query is not None else len(thread_query))
        nq = T + 2
        kmax = max(self.TERMS_K, self.MESSAGES_K, self.THREADS_K)
        if self._pinned_q is None or self._pinned_q.shape != (nq, dim):
            dev = torch.device("cuda", self.device)
            self._pinned_q = torch.empty((nq, dim), dtype=torch.float32).pin_memory()
            self._dev_q = torch.empty((nq, dim), dtype=torch.float32, device=dev)
            # the result keys land in pinned host memory straight from the last kernel of every lookup (no device buffer, no copy back, no
            # memset launch in front: at the reference's scale a user query is launch-bound -- ten submissions were 98 us, profiles/r06_raw/
            # cfg5_reference_scale.txt); the views handed to the engine are made once per shape (a tensor slice costs 2-3 us of host time)
            self._pinned_keys = torch.zeros((nq, kmax), dtype=torch.int64).pin_memory()
            self._keys_np = self._pinned_keys.numpy()
            self._host_q = self._pinned_q.numpy()
            self._views = {
                "tq": self._dev_q[:T], "tk": self._pinned_keys[:T, : self.TERMS_K],
                "mq": self._dev_q[T : T + 1], "mq1": self._dev_q[T], "mk": self._pinned_keys[T : T + 1, : self.MESSAGES_K],
                "hq": self._dev_q[T + 1 : T + 2], "hk": self._pinned_keys[T + 1 : T + 2, : self.THREADS_K],
            }
        host_q, views, keys_np = self._host_q, self._views, self._keys_np
        if T:
            host_q[:T] = tq
        host_q[T] = 0 if message_query is None else np.asarray(message_query, dtype=np.float32)
        host_q[T + 1] = 0 if thread_query is None else np.asarray(thread_query, dtype=np.float32)
        thr = _native.f32_threshold
        subset = None
        ran_t = ran_m = ran_h = False
        with torch.cuda.stream(self.stream):
            self._dev_q.copy_(self._pinned_q, non_blocking=True)
            if T and self._use("terms"):
                self.engine.search_device(views["tq"], self.TERMS_K, float(thr(self.TERMS_MIN)), out_keys=views["tk"])
                ran_t = True
            if message_query is not None and self._use("messages"):
                if message_subset is not None:
                    subset = np.asarray(message_subset, dtype=np.int64).reshape(-1)
                    n = self.corpora["messages"][1]
                    rows = np.where(subset < 0, subset + n, subset)
                    if ((rows < 0) | (rows >= n)).any():
                        raise IndexError("message subset ordinal out of range")
                    if len(rows):
                        d_rows = torch.from_numpy(rows.astype(np.int32)).to(self._dev_q.device, non_blocking=True)
                        self.engine.search_subset_device(views["mq1"], d_rows, self.MESSAGES_K, float(thr(self.MESSAGES_MIN)), out_keys=views["mk"])
                        ran_m = True
                else:
                    self.engine.search_device(views["mq"], self.MESSAGES_K, float(thr(self.MESSAGES_MIN)), out_keys=views["mk"])
                    ran_m = True
            if thread_query is not None and self._use("threads"):
                self.engine.search_device(views["hq"], self.THREADS_K, float(thr(self.THREADS_MIN)), out_keys=views["hk"])
                ran_h = True
        self.stream.synchronize()
        # (a lookup that did not run leaves last call's keys in its rows: empty them; the others were overwritten in full by their merge kernels)
        if T and not ran_t:
            keys_np[:T] = 0
        if not ran_m:
            keys_np[T] = 0
        if not ran_h:
            keys_np[T + 1] = 0
        if T and self.TERMS_K < kmax:
            keys_np[:T, self.TERMS_K :] = 0
        keys_np[T, self.MESSAGES_K :] = 0
        keys_np[T + 1, self.THREADS_K :] = 0
        ords, scs, cnts = _native.decode_keys(keys_np)

        def hits(row: int, remap=None) -> list[ScoredInt]:
            m = int(cnts[row])
            items = ords[row, :m].tolist()
            if remap is not None:
                items = [int(remap[i]) for i in items]
            return [ScoredInt(int(i), float(s)) for i, s in zip(items, scs[row, :m].tolist())]

        return FusedResult([hits(i) for i in range(T)], hits(T, subset), hits(T + 1))
