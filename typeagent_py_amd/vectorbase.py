"""Drop-in for `typeagent.aitools.vectorbase` with the nearest-neighbour search
running on MI355X (gfx950) through libtavb.so.

API mirror of the reference's `src/typeagent/aitools/vectorbase.py`
(/root/reference; line numbers below refer to it):

  DEFAULT_MIN_SCORE, MODEL_DEFAULT_MIN_SCORES, get_default_min_score   :16-41
  cosine_to_score                                                      :44-47
  ScoredInt                                                            :50-55
  TextEmbeddingIndexSettings                                           :58-79
  VectorBase                                                           :82-287

Same names, arguments, defaults, return types and exceptions.  What differs is
where the arithmetic runs: `fuzzy_lookup_embedding*` launch HIP kernels (fused
dot + score + threshold + top-k) instead of np.dot/argpartition, and there is
one additive method, `fuzzy_lookup_embeddings` (a batch of queries in one
submission).  The host ndarray `_vectors` stays the authoritative copy for
serialize()/deserialize(); the device buffer mirrors it and is synced lazily,
appends moving only the new rows.

There is no CPU fallback: a lookup on a non-empty index without libtavb.so or
without a visible MI355X raises RuntimeError.

Documented deviations from the reference (none is exercised by its callers):
  * equal float32 scores are ordered by ascending ordinal (the reference's order
    among ties is whatever numpy's introselect leaves, :183-187);
  * a query is converted to float32 (np.dot would promote a float64 query);
  * negative `max_hits` raises ValueError (the reference returns slicing artefacts).
"""

from __future__ import annotations

import gc
import os
import time
from collections.abc import Callable
from dataclasses import dataclass

import numpy as np

from . import _native
from .embeddings import IEmbeddingModel, NormalizedEmbedding, NormalizedEmbeddings

DEFAULT_MIN_SCORE = 0.85

# Per-model score cut-offs the reference ships for the built-in OpenAI models (:31-35).
MODEL_DEFAULT_MIN_SCORES: dict[str, float] = {
    "text-embedding-3-large": 0.74,
    "text-embedding-3-small": 0.73,
    "text-embedding-ada-002": 0.93,
}

_PAGE = _native.MAX_FUSED_K  # most hits the fused select-while-streaming kernels return; beyond: one emit-all pass + host sort


try:  # whole-matrix digests of the "full" host watch
    from xxhash import xxh3_128_digest as _digest
except Exception:  # pragma: no cover - xxhash is part of the image
    import hashlib

    def _digest(data) -> bytes:
        return hashlib.blake2b(data, digest_size=16).digest()


class _Watch:
    """Dirty flag shared by an index and every view of its host matrix it has handed out (and by the indexes that adopted such a
    view through deserialize(): `followers`)."""

    __slots__ = ("dirty", "followers")

    def __init__(self) -> None:
        self.dirty = False
        self.followers: list[_Watch] = []

    def touch(self) -> None:
        self.dirty = True
        for f in self.followers:
            f.dirty = True


class _WatchedMatrix(np.ndarray):
    """What serialize() / `_vectors` / get_embedding_at() hand out for a matrix this index owns: a plain float32 view of the live
    host matrix (the reference hands out the live `_vectors`, vectorbase.py:268-271) that remembers being written to.  The
    reference always scores the live matrix (:176); here the device mirror is refreshed on the next lookup after a write made
    through numpy's array API on this array or on a view derived from it: item / slice assignment, in-place operators and ufunc
    `out=`, fill / sort / put / ..., np.copyto / np.put / np.place / np.putmask, `out=` / `dst=` of any other numpy function, `.flat`
    (touching `.flat` counts as a write: the iterator it returns assigns behind numpy's back).
    Arrays that merely DERIVE from the matrix but own their memory (m.copy(), m * 2, a matrix product with it) are plain results: writing to them
    does not mark anything.  What this class cannot see -- a base-class view from np.asarray(), memoryview, ctypes pointers, torch.from_numpy,
    another library writing through the buffer protocol -- is caught by the fingerprint the index keeps of a matrix it has handed out
    (`verify_host`: sampled rows by default, so a bulk rewrite is noticed, a single-row edit by such a route is not): those writers
    call `VectorBase.mark_dirty()`."""

    _tavb_watch: _Watch | None = None

    def __array_finalize__(self, obj) -> None:
        # only arrays that SHARE the matrix' memory inherit the watch (a copy that owns its data is nobody's mirror: editing it must not
        # trigger a re-upload of a multi-GB corpus)
        if obj is not None and not self.flags.owndata:
            self._tavb_watch = getattr(obj, "_tavb_watch", None)

    def _touch(self) -> None:
        w = self._tavb_watch
        if w is not None:
            w.touch()

    def __setitem__(self, key, value) -> None:
        self._touch()
        super().__setitem__(key, value)

    @property
    def flat(self):
        self._touch()  # `m.flat[i] = x` assigns through a np.flatiter: conservative, also for reads
        return np.ndarray.flat.__get__(self)

    @flat.setter
    def flat(self, value) -> None:
        self._touch()
        np.ndarray.flat.__set__(self, value)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        plain = lambda a: a.view(np.ndarray) if isinstance(a, _WatchedMatrix) else a
        if out is not None:
            for o in out:
                if isinstance(o, _WatchedMatrix):
                    o._touch()
            kwargs["out"] = tuple(plain(o) for o in out)
        if method == "at" and isinstance(inputs[0], _WatchedMatrix):
            inputs[0]._touch()
        result = getattr(ufunc, method)(*(plain(i) for i in inputs), **kwargs)
        if out is not None and method == "__call__" and len(out) == 1:
            return out[0]  # `m *= 2` must rebind m to itself
        return result  # plain ndarrays: results of arithmetic on the matrix are copies, not watched

    def __array_function__(self, func, types, args, kwargs):
        if func in _WRITING_FUNCTIONS and args and isinstance(args[0], _WatchedMatrix):
            args[0]._touch()
        for name in ("out", "dst", "a"):  # np.take(..., out=m), a product written into out=m, np.copyto(dst=m, src=...), np.put(a=m, ...)
            target = kwargs.get(name)
            if name == "a" and func not in _WRITING_FUNCTIONS:
                continue
            for t in target if isinstance(target, tuple) else (target,):
                if isinstance(t, _WatchedMatrix):
                    t._touch()
        result = super().__array_function__(func, types, args, kwargs)
        if isinstance(result, _WatchedMatrix) and result.flags.owndata:
            return result.view(np.ndarray)  # a fresh array (a product, a sorted copy, ...): nobody's mirror
        return result

    def _writing_method(name):  # noqa: N805 -- class-body helper
        base = getattr(np.ndarray, name)

        def method(self, *args, **kwargs):
            self._touch()
            return base(self, *args, **kwargs)

        method.__name__ = name
        return method

    fill = _writing_method("fill")
    sort = _writing_method("sort")
    partition = _writing_method("partition")
    put = _writing_method("put")
    setfield = _writing_method("setfield")
    byteswap = _writing_method("byteswap")
    del _writing_method


_WRITING_FUNCTIONS = {np.copyto, np.put, np.place, np.putmask, np.put_along_axis, np.fill_diagonal}


def get_default_min_score(model_name: str) -> float:
    return MODEL_DEFAULT_MIN_SCORES.get(model_name, DEFAULT_MIN_SCORE)


def cosine_to_score(cosine_similarity: np.ndarray) -> np.ndarray:
    """Host-side statement of the score map the kernels apply per row: cosine in
    [-1, 1] -> public score in [0, 1]."""
    return np.clip((cosine_similarity + 1.0) / 2.0, 0.0, 1.0)


@dataclass(slots=True)  # same fields / equality / repr as the reference's (vectorbase.py:50-55); slots halve the cost of building 32k hits per 1024-query batch
class ScoredInt:
    item: int
    score: float


def _default_embedding_model() -> IEmbeddingModel:
    # The reference builds its default provider model here (:74, model_adapters
    # .create_embedding_model).  That is LLM/HTTP plumbing outside this package: use the
    # reference's factory when typeagent is importable, otherwise insist on an explicit model.
    try:
        from typeagent.aitools.model_adapters import create_embedding_model  # type: ignore
    except Exception as exc:  # pragma: no cover - depends on the host environment
        raise RuntimeError(
            "TextEmbeddingIndexSettings needs an explicit embedding_model when typeagent's "
            "model_adapters is not importable"
        ) from exc
    return create_embedding_model()


@dataclass
class TextEmbeddingIndexSettings:
    embedding_model: IEmbeddingModel
    min_score: float  # 0..1
    max_matches: int | None  # >= 1, None = no limit
    batch_size: int  # >= 1

    def __init__(
        self,
        embedding_model: IEmbeddingModel | None = None,
        min_score: float | None = None,
        max_matches: int | None = None,
        batch_size: int | None = None,
    ):
        self.embedding_model = embedding_model or _default_embedding_model()
        name = getattr(self.embedding_model, "model_name", "")
        self.min_score = get_default_min_score(name) if min_score is None else min_score
        self.max_matches = max_matches if (max_matches and max_matches >= 1) else None
        self.batch_size = batch_size if (batch_size and batch_size >= 1) else 8


def _env_dtype() -> int:
    v = os.environ.get("TYPEAGENT_VB_DTYPE", "fp32").lower()
    if v in ("fp16", "f16", "half"):
        return _native.TAVB_F16
    return _native.TAVB_F32


try:  # the C builder of the hit lists (csrc/tavb_pyhits.c, built next to libtavb.so); host-side only -- the Python loop below does the same work
    from . import _tavb_pyhits
except ImportError:  # pragma: no cover - the module ships with the package
    _tavb_pyhits = None


def _scored_lists(ords: np.ndarray, scs: np.ndarray, cnts: np.ndarray, width: int) -> list[list[ScoredInt]]:
    """[Q, width] result arrays -> Q lists of ScoredInt (what vectorbase.py:188-190 builds per query).  A 1024 x 32 batch is 32k
    Python objects: the cyclic collector would wake ~45 times while they are allocated (none of them can be part of a cycle: an
    int and a float each) -- it is paused for the duration when it was on and the batch is big.  The objects are built in C
    (`_tavb_pyhits.build`: tp_alloc + two slot stores per hit, ~45 ns against ~110 ns through the dataclass __init__)."""
    pause = len(cnts) * width >= 4096 and gc.isenabled()
    if pause:
        gc.disable()
    try:
        if _tavb_pyhits is not None and ords.dtype == np.int64 and scs.dtype == np.float32 and cnts.dtype == np.int32 \
                and ords.flags.c_contiguous and scs.flags.c_contiguous and cnts.flags.c_contiguous and ords.shape == scs.shape == (len(cnts), width):
            return _tavb_pyhits.build(ScoredInt, ords, scs, cnts, width)
        rows_o, rows_s = ords.tolist(), scs.tolist()
        counts = cnts.tolist()
        return [list(map(ScoredInt, o, s_)) if m == width else list(map(ScoredInt, o[:m], s_[:m])) for o, s_, m in zip(rows_o, rows_s, counts)]
    finally:
        if pause:
            gc.enable()


class VectorBase:
    settings: TextEmbeddingIndexSettings
    _model: IEmbeddingModel
    _embedding_size: int

    def __init__(
        self,
        settings: TextEmbeddingIndexSettings,
        *,
        device: int | None = None,
        corpus_dtype: str | None = None,
        devices: list[int] | None = None,
        keep_host_copy: bool | None = None,
        verify_host: str | None = None,
    ):
        self.settings = settings
        self._model = settings.embedding_model
        self._embedding_size = 0
        self._device_index = device if device is not None else (int(os.environ["TYPEAGENT_VB_DEVICE"]) if "TYPEAGENT_VB_DEVICE" in os.environ else None)
        # several GPUs under ONE object (row shards, one context + stream per device; typeagent_py_amd/multidevice.py)
        if devices is None and os.environ.get("TYPEAGENT_VB_DEVICES"):
            devices = [int(x) for x in os.environ["TYPEAGENT_VB_DEVICES"].split(",") if x.strip() != ""]
        self._devices = list(devices) if devices else None
        # keep_host_copy=False: rows handed to add_embedding(s) are streamed straight into the device corpus (pinned
        # staging, async H2D, on-device fp16 conversion) and NOT kept on the host -- the load paths
        # (adapters.load_embeddings_bin / load_sqlite_embeddings) for corpora that should not exist twice.  serialize() /
        # get_embedding_at() copy the matrix back on first use, after which the host copy is authoritative again.
        if keep_host_copy is None:
            keep_host_copy = os.environ.get("TYPEAGENT_VB_HOST_COPY", "1") not in ("0", "false", "no")
        self._keep_host = bool(keep_host_copy) or bool(self._devices)
        if corpus_dtype is None:
            self._dtype = _env_dtype()
        else:
            self._dtype = _native.TAVB_F16 if corpus_dtype.lower() in ("fp16", "f16", "half", "float16") else _native.TAVB_F32
        self._engine: _native.Engine | None = None
        self._host = np.zeros((0,), dtype=np.float32)  # backing store; _vectors is a view of its first _count rows
        self._count = 0
        self._dev_rows = 0  # rows of the host matrix already mirrored on the device
        self._dev_valid = True  # False => the device copy must be rebuilt from row 0
        # The live host matrix can be edited in place by whoever holds it (the reference always scores the live matrix, :176):
        #  * a matrix this index OWNS is handed out as a _WatchedMatrix view (`_watch` is its dirty flag);
        #  * a matrix the CALLER owns (deserialize(data) keeps `data` by reference, :287) cannot be wrapped: `_handed_out` turns on a
        #    fingerprint check per lookup -- "sampled" (default: 32 rows, ~30 us) or "full" (every byte: TYPEAGENT_VB_VERIFY_HOST=full /
        #    verify_host="full"; ~0.1 ms per MB) -- and mark_dirty() is the explicit form.
        self._watch = _Watch()
        self._view = None  # (host buffer, row count, the write-tracking view handed out for them)
        self._view_out = False  # a view of a matrix this index owns has been handed out: fingerprint fallback on (round-3 advice)
        self._handed_out = False
        self._dev_fingerprint = None
        self._fp_skipped = 0      # lookups since the fallback fingerprint of an owned, handed-out matrix was last compared
        self._fp_checked = 0.0    # ... and when (time.monotonic())
        mode = (verify_host or os.environ.get("TYPEAGENT_VB_VERIFY_HOST", "sampled")).lower()
        if mode not in ("sampled", "lazy", "full", "off"):
            raise ValueError("verify_host must be 'sampled', 'lazy', 'full' or 'off'")
        self._verify_host = mode
        self._device_only = None  # torch tensor when the corpus lives only on the device
        self._subset_cache = None  # (caller's subset object, private copy, row count, ordinals int64, device int32 rows) of the last subset lookup
        self._row_messages: np.ndarray | None = None  # chunk row -> message ordinal (message re-rank on the device)
        self._row_messages_rows = -1  # rows of the map already on the device
        self.clear()

    # ------------------------------------------------------------------ storage
    @property
    def _vectors(self) -> NormalizedEmbeddings:
        if self._device_only is not None:
            self._materialize_host()
        if self._handed_out:
            return self._host  # the caller's own matrix, adopted by reference: the same object every time (:271, :287)
        live = self._host
        cached = self._view
        if cached is not None and cached[0] is live and cached[1] == self._count:
            return cached[2]  # the same object for as long as the matrix is the same (the reference returns its one `_vectors`)
        if self._embedding_size > 0 and live.ndim == 2 and self._count != live.shape[0]:
            view = live[: self._count].view(_WatchedMatrix)  # the filled part of the growth buffer
        else:
            view = live.view(_WatchedMatrix)
        view._tavb_watch = self._watch
        self._view = (live, self._count, view)
        if not self._view_out:
            # from now on the lookups also keep the fingerprint of this matrix (writers the view cannot see); nobody has held a
            # view until now, so the mirror -- if there is one -- matches the matrix as it is
            self._view_out = True
            if self._dev_valid and self._dev_rows == self._count:
                self._dev_fingerprint = self._fingerprint()
        return view

    @_vectors.setter
    def _vectors(self, value: NormalizedEmbeddings) -> None:
        self._adopt_host(value)

    def _adopt_host(self, matrix: np.ndarray) -> None:
        self._device_only = None
        self._host = matrix
        self._count = len(matrix)
        self._dev_rows = 0
        self._dev_valid = False
        self._handed_out = True  # the caller keeps a reference to `matrix` (deserialize keeps it by reference, :287)
        other = getattr(matrix, "_tavb_watch", None)
        if other is not None and other is not self._watch and self._watch not in other.followers:
            other.followers.append(self._watch)  # another index's serialize() output: writes through it reach this index too

    def _materialize_host(self) -> None:
        t, n = self._device_only, self._count
        self._device_only = None
        if isinstance(t, list):
            host = np.concatenate([x.float().cpu().numpy() for x in t])[:n]
        else:
            host = t[:n].float().cpu().numpy()
        self._keep_host = True  # from here on the host matrix is the authoritative copy (like the reference's)
        self._host, self._count = host, n  # device copy stays valid: same rows
        if self._dtype == _native.TAVB_F16:
            pass  # host copy holds the fp16 values widened to f32

    def _reserve(self, extra: int) -> None:
        need = self._count + extra
        if self._host.ndim != 2 or self._host.shape[1] != self._embedding_size:
            self._host = np.zeros((max(need, 4), self._embedding_size), dtype=np.float32)
            self._handed_out = self._view_out = False
            return
        if need > self._host.shape[0] or self._handed_out:
            # (an adopted matrix is the caller's: appends go to a buffer of our own, like the reference's np.append copy, :128)
            if (self._handed_out or self._view_out) and self._dev_valid and self._dev_rows == self._count and self._dev_fingerprint != self._fingerprint():
                self._dev_valid = False  # edited in place since the last upload (also through a view the tracker cannot see): the rows already mirrored are stale too
            grown = np.empty((max(need, 2 * self._host.shape[0], 4), self._embedding_size), dtype=np.float32)
            grown[: self._count] = self._host[: self._count]
            self._host = grown
            self._handed_out = self._view_out = False  # (views of the old buffer are no longer views of this index' matrix)

    async def get_embedding(self, key: str, cache: bool = True) -> NormalizedEmbedding:
        if cache:
            return await self._model.get_embedding(key)
        return await self._model.get_embedding_nocache(key)

    async def get_embeddings(self, keys: list[str], cache: bool = True) -> NormalizedEmbeddings:
        if cache:
            return await self._model.get_embeddings(keys)
        return await self._model.get_embeddings_nocache(keys)

    def __len__(self) -> int:
        return self._count

    def __bool__(self) -> bool:  # an empty index is still truthy (:111-113)
        return True

    def add_embedding(self, key: str | None, embedding: NormalizedEmbedding | list[float]) -> None:
        row = np.asarray(embedding, dtype=np.float32)
        if self._embedding_size == 0:
            self._set_embedding_size(len(row))
        if len(row) != self._embedding_size:
            raise ValueError(f"Embedding size mismatch: expected {self._embedding_size}, got {len(row)}")
        if self._stream_to_device(row.reshape(1, -1)):
            pass
        else:
            if self._device_only is not None:
                self._materialize_host()
            self._reserve(1)
            self._host[self._count] = row.reshape(-1)
            self._count += 1
        if key is not None:
            self._model.add_embedding(key, row)

    def add_embeddings(self, keys: None | list[str], embeddings: NormalizedEmbeddings) -> None:
        if embeddings.ndim != 2:
            raise ValueError(f"Expected 2D embeddings array, got {embeddings.ndim}D")
        if self._embedding_size == 0:
            self._set_embedding_size(embeddings.shape[1])
        if embeddings.shape[1] != self._embedding_size:
            raise ValueError(f"Embedding size mismatch: expected {self._embedding_size}, got {embeddings.shape[1]}")
        if not self._stream_to_device(embeddings):
            if self._device_only is not None:
                self._materialize_host()
            n = embeddings.shape[0]
            self._reserve(n)
            self._host[self._count : self._count + n] = embeddings
            self._count += n
        if keys is not None:
            for key, row in zip(keys, embeddings):
                self._model.add_embedding(key, row)

    def _stream_to_device(self, rows: np.ndarray) -> bool:
        """keep_host_copy=False: append `rows` to the device corpus only.  False when this index keeps its host matrix."""
        if self._keep_host or (self._count > 0 and self._device_only is None):
            return False  # (a host matrix already exists -- e.g. after serialize(): stay host-authoritative)
        eng = self._ensure_engine()
        if len(rows):
            eng.upload_rows(np.ascontiguousarray(rows, dtype=np.float32), self._count, self._dtype)
            self._count += len(rows)
            self._device_only = eng.corpus
            self._dev_rows, self._dev_valid = self._count, True
        return True

    async def add_key(self, key: str, cache: bool = True) -> None:
        embedding = await self.get_embedding(key, cache=cache)
        self.add_embedding(key if cache else None, embedding)

    async def add_keys(self, keys: list[str], cache: bool = True) -> NormalizedEmbeddings | None:
        if not keys:
            return None
        embeddings = await self.get_embeddings(keys, cache=cache)
        self.add_embeddings(keys if cache else None, embeddings)
        return embeddings

    # ------------------------------------------------------------------ device mirror
    def _ensure_engine(self) -> _native.Engine:
        if self._engine is None:
            if self._devices:
                from .multidevice import DeviceGroup

                self._engine = DeviceGroup(self._devices)
            else:
                self._engine = _native.Engine(self._device_index)
            level = os.environ.get("TYPEAGENT_VB_F32_SHADOW", "1").lower()
            if level in ("0", "false", "no"):
                self._engine.set_option("f32_shadow", 0)  # no fp16 shadow of fp32 corpora (+50 % device memory) for large batches
            elif level == "2":
                self._engine.set_option("f32_shadow", 2)  # every lookup on big fp32 corpora filters on the shadow (tavb.h)
        return self._engine

    def _sync_device(self) -> _native.Engine:
        eng = self._ensure_engine()
        if self._device_only is not None:
            return eng
        n = self._count
        if self._watch.dirty:  # a view handed out by serialize() / _vectors / get_embedding_at() was written to
            self._watch.dirty = False
            self._dev_valid = False
        if (self._handed_out or self._view_out) and self._dev_valid and self._dev_rows == n and n > 0 and self._fingerprint_due() \
                and self._dev_fingerprint != self._fingerprint():
            self._dev_valid = False  # the caller's matrix adopted by deserialize() -- or a matrix of ours somebody holds a view of -- was edited in place
        if not self._dev_valid:
            self._dev_rows = 0
            self._dev_valid = True
        if self._dev_rows > n:
            self._dev_rows = 0
        if self._dev_rows < n or eng.rows != n or eng.dim != self._embedding_size:
            start = self._dev_rows if (eng.corpus is not None and eng.dim == self._embedding_size and eng.dtype == self._dtype) else 0
            done = eng.upload_rows(self._host[start:n], start, self._dtype, capacity_hint=self._host.shape[0] if start == 0 else 0)
            if done is False:  # a device group has to re-shard: everything again
                eng.upload_rows(self._host[:n], 0, self._dtype, capacity_hint=self._host.shape[0])
            self._dev_rows = n
            self._dev_fingerprint = self._fingerprint() if (self._handed_out or self._view_out) else None
        return eng

    def _fingerprint_due(self) -> bool:
        """A matrix somebody else can write to -- the CALLER's own (deserialize(), :287), or one of ours that has been handed out as a view
        (serialize() / _vectors / get_embedding_at()) -- is fingerprinted before EVERY lookup (the reference always scores the live matrix,
        :176).  The view reports writers that use numpy's array API by itself; the fingerprint is what notices the others
        (torch.from_numpy on the view, ctypes, a C extension).  It costs as much as a whole lookup on a small corpus (~30 us), so
        `verify_host="lazy"` (opt-in) takes it for a matrix of ours at most once per 64 lookups or 20 ms, whichever comes first --
        such a writer may then be served stale answers for that long; `mark_dirty()` is the explicit, immediate form."""
        if self._handed_out or self._verify_host != "lazy":
            return True
        self._fp_skipped += 1
        now = time.monotonic()
        if self._fp_skipped >= 64 or now - self._fp_checked >= 0.02:
            self._fp_skipped = 0
            self._fp_checked = now
            return True
        return False

    def _fingerprint(self):
        """Watch on a matrix the CALLER owns (adopted by deserialize(), :287).  "sampled": hash of up to 32 evenly spaced rows
        (~30 us): whole-matrix edits (re-normalisation, bulk replacement) are caught, an edit confined to rows outside the
        sample is not.  "full": a 128-bit hash of every byte (xxh3, ~0.1 ms per MB on one core; blake2b when xxhash is not
        installed): nothing is missed, at a per-lookup cost that only small indexes can afford.  `mark_dirty()` is the explicit form."""
        n = self._count
        if n == 0 or self._host.ndim != 2 or self._verify_host == "off":
            return None
        if self._verify_host == "full":  # ("lazy" fingerprints like "sampled", less often: _fingerprint_due)
            data = memoryview(np.ascontiguousarray(self._host[:n])).cast("B")
            return (n, self._host.shape[1], _digest(data))
        idx = np.unique(np.linspace(0, n - 1, num=min(n, 32)).astype(np.int64))
        return hash((n, self._host.shape[1], self._host[idx].tobytes()))

    def mark_dirty(self) -> None:
        """Call after writing to the host matrix by a route numpy does not see (raw pointers, memoryview, a base-class view from
        np.asarray(serialize())), or -- for a matrix adopted by deserialize() under the default sampled watch -- after editing
        single rows of it: the device mirror is rebuilt on the next lookup."""
        self._dev_valid = False

    def adopt_device_corpus(self, tensor, rows: int | None = None, ordinal_base: int = 0) -> None:
        """Use a float32/float16 torch tensor [N, D] already on the GPU as the corpus
        without a host copy (corpora larger than host RAM).  serialize() will copy
        it back on demand.  With `devices=[...]`: a list of tensors, one row shard per device."""
        eng = self._ensure_engine()
        eng.ordinal_base = ordinal_base
        if self._devices:  # one tensor per device, row shards in order
            tensors = list(tensor) if isinstance(tensor, (list, tuple)) else [tensor]
            eng.set_shard_tensors(tensors, ordinal_base=ordinal_base)
            tensor = tensors
            self._set_embedding_size(int(tensors[0].shape[1]))
        else:
            eng.set_corpus_tensor(tensor, rows=rows, ordinal_base=ordinal_base)
            self._set_embedding_size(int(tensor.shape[1]))
        self._device_only = tensor
        self._count = eng.rows
        self._dtype = eng.dtype
        self._dev_rows, self._dev_valid = eng.rows, True
        self._host = np.zeros((0, self._embedding_size), dtype=np.float32)

    @property
    def engine(self) -> _native.Engine:
        """The device engine with the corpus synced (tuning knobs, profiling, device-resident calls)."""
        return self._sync_device()

    # ------------------------------------------------------------------ lookups
    @staticmethod
    def _limits(max_hits: int | None, min_score: float | None) -> tuple[int, np.float32]:
        if max_hits is None:
            max_hits = 10
        if min_score is None:
            min_score = 0.0
        if max_hits < 0:
            raise ValueError("max_hits must be >= 0")
        return max_hits, _native.f32_threshold(min_score)

    def fuzzy_lookup_embedding(
        self,
        embedding: NormalizedEmbedding,
        max_hits: int | None = None,
        min_score: float | None = None,
        predicate: Callable[[int], bool] | None = None,
    ) -> list[ScoredInt]:
        max_hits, thr = self._limits(max_hits, min_score)
        if self._count == 0:
            return []
        eng = self._sync_device()
        if predicate is None:
            if 1 <= max_hits <= _PAGE:
                ids, scs = eng.search(embedding, max_hits, thr)  # fused select-while-streaming
            else:
                # more hits than the fused selection holds, or max_hits == 0 (every survivor, sorted: the `[-0:]` quirk,
                # :186-187): ONE pass emits all survivors, the host sorts them
                ids, scs = eng.search_all(embedding, thr, None if max_hits == 0 else max_hits)
            return list(map(ScoredInt, ids.tolist(), scs.tolist()))  # (tolist() yields Python ints / floats: the float32 scores widened, as the reference's float(score))
        # predicate path (:191-201): threshold on the device (one pass, all survivors), then exactly the reference's steps:
        # predicate(ordinal) for EVERY survivor in ascending ordinal order, stable sort by score, cut
        ids, scs = eng.search_all(embedding, thr, None)
        order = np.lexsort((ids,))  # ascending ordinal: the order of np.flatnonzero (:193)
        kept = [ScoredInt(int(i), float(s)) for i, s in zip(ids[order].tolist(), scs[order].tolist()) if predicate(int(i))]
        kept.sort(key=lambda x: x.score, reverse=True)
        return kept[:max_hits]

    def fuzzy_lookup_embedding_in_subset(
        self,
        embedding: NormalizedEmbedding,
        ordinals_of_subset: list[int],
        max_hits: int | None = None,
        min_score: float | None = None,
    ) -> list[ScoredInt]:
        max_hits, thr = self._limits(max_hits, min_score)
        if len(ordinals_of_subset) == 0 or self._count == 0:
            return []
        n = self._count
        # The same subset again (the memory provider hands in one scope list per query term, storage/memory/messageindex.py:173-183;
        # tools/benchmark_vectorbase.py:133-163 one list for every round): its wrapped, range-checked row list stays on the device.
        # Recognised by identity AND content -- `==` on the caller's list against a private copy (identical int objects compare by
        # pointer: ~3 ns per ordinal, against ~30 ns to convert one), so a list edited in place since is simply a new subset.
        cached = self._subset_cache
        hit = (cached is not None and cached[0] is ordinals_of_subset and cached[2] == n and len(cached[1]) == len(ordinals_of_subset)
               and (np.array_equal(cached[1], ordinals_of_subset) if isinstance(ordinals_of_subset, np.ndarray) else cached[1] == ordinals_of_subset))
        if hit:
            subset, rows, dev_rows = cached[3], None, cached[4]
        else:
            subset = np.asarray(ordinals_of_subset)
            if subset.dtype.kind not in "iu":
                raise IndexError("arrays used as indices must be of integer (or boolean) type")
            subset = subset.astype(np.int64, copy=False).reshape(-1)
            rows = np.where(subset < 0, subset + n, subset)  # numpy index wrap (:218)
            bad = (rows < 0) | (rows >= n)
            if bad.any():
                first = int(subset[np.argmax(bad)])
                raise IndexError(f"index {first} is out of bounds for axis 0 with size {n}")
            dev_rows = None
        eng = self._sync_device()
        if 1 <= max_hits <= _PAGE and isinstance(eng, _native.Engine) and isinstance(ordinals_of_subset, (list, np.ndarray)):
            if dev_rows is None:
                dev_rows = eng.rows_to_device(rows)
                is_array = isinstance(ordinals_of_subset, np.ndarray)  # (then `subset` may be a view of the caller's array: keep a copy)
                keep = ordinals_of_subset.copy() if is_array else list(ordinals_of_subset)
                subset = subset.copy() if is_array else subset
                self._subset_cache = (ordinals_of_subset, keep, n, subset, dev_rows)
            pos, scs = eng.search_subset_resident(embedding, dev_rows, max_hits, thr)
        else:
            if rows is None:
                rows = np.where(subset < 0, subset + n, subset)
            if 1 <= max_hits <= _PAGE:
                pos, scs = eng.search_subset(embedding, rows, max_hits, thr)
            else:
                pos, scs = eng.search_all(embedding, thr, None if max_hits == 0 else max_hits, subset_rows=rows)
        return list(map(ScoredInt, subset[pos].tolist(), scs.tolist()))  # (the caller's ordinals at the returned positions, :229)

    def fuzzy_lookup_embeddings(
        self,
        embeddings: NormalizedEmbeddings,
        max_hits: int | None = None,
        min_score: float | None = None,
        as_arrays: bool = False,
    ):
        """Batch form: equals [fuzzy_lookup_embedding(e, max_hits, min_score) for e in embeddings],
        served by one device submission (the reference loops, storage/memory/reltermsindex.py:320-332).
        `min_score` may be a sequence with one threshold per query.
        `as_arrays=True` (1 <= max_hits <= 256) returns (ordinals int64 [Q, max_hits], scores float32 [Q, max_hits], counts
        int32 [Q]) instead of Q lists of ScoredInt: building 32k Python objects takes as long as a third of the 1024-query
        lookup over 10M rows itself."""
        queries = np.asarray(embeddings, dtype=np.float32)
        if queries.ndim != 2:
            raise ValueError(f"Expected 2D embeddings array, got {queries.ndim}D")
        if min_score is not None and not np.isscalar(min_score) and np.ndim(min_score) == 1:
            # one threshold per query (Q calls of the reference have Q `min_score` arguments, :163-173): same kernels as a uniform batch
            if len(min_score) != len(queries):
                raise ValueError(f"Number of thresholds {len(min_score)} does not match number of embeddings {len(queries)}")
            per_query = list(min_score)
            max_hits, _ = self._limits(max_hits, 0.0)
            thr = np.asarray([_native.f32_threshold(0.0 if m is None else m) for m in per_query], dtype=np.float32)
            min_score = None
        else:
            per_query = None
            max_hits, thr = self._limits(max_hits, min_score)
        if as_arrays and not (1 <= max_hits <= _PAGE):
            raise ValueError(f"as_arrays needs 1 <= max_hits <= {_PAGE}")
        if self._count == 0 or len(queries) == 0:
            if as_arrays:
                nq = len(queries)
                return np.zeros((nq, max_hits), np.int64), np.zeros((nq, max_hits), np.float32), np.zeros(nq, np.int32)
            return [[] for _ in range(len(queries))]
        if not (1 <= max_hits <= _PAGE):
            return [self.fuzzy_lookup_embedding(q, max_hits, min_score if per_query is None else per_query[i]) for i, q in enumerate(queries)]
        eng = self._sync_device()
        ords, scs, cnts = eng.search_batch(queries, max_hits, thr)
        if as_arrays:
            return ords, scs, cnts
        return _scored_lists(ords, scs, cnts, max_hits)

    # ------------------------------------------------------------------ message re-rank (additive)
    def set_row_messages(self, row_to_message) -> None:
        """chunk row -> message ordinal for every row of the index (-1: none): what the providers keep as
        `TextLocation.message_ordinal` per index position (knowpro/textlocindex.py:54-73; the `msg_id` column of
        storage/sqlite/schema.py:71-81).  Enables `lookup_messages_by_embedding*`, which run the providers' post-lookup
        aggregation on the device."""
        self._row_messages_src = row_to_message  # identity of the caller's object: callers that pass it every time do not re-upload
        self._row_messages = np.array(row_to_message, dtype=np.int64).reshape(-1)  # a snapshot: never an alias of the caller's array
        self._row_messages_rows = -1  # (re-)uploaded on the next message lookup

    def _messages_engine(self):
        """engine with corpus AND map synced, or None when the aggregation has to run on the host (device group)."""
        if self._row_messages is None:
            raise RuntimeError("set_row_messages() first")
        if len(self._row_messages) < self._count:
            raise ValueError(f"the row -> message map covers {len(self._row_messages)} rows, the index has {self._count}")
        eng = self._sync_device()
        if not isinstance(eng, _native.Engine):
            return None
        if self._row_messages_rows != self._count:
            eng.set_row_messages(self._row_messages[: self._count])
            self._row_messages_rows = self._count
        return eng

    def _host_rerank(self, hits, max_matches, accept=None) -> list[ScoredInt]:
        best: dict[int, float] = {}
        for h in hits:
            msg = int(self._row_messages[h.item])
            if msg < 0 or (accept is not None and msg not in accept):
                continue
            if msg not in best:
                best[msg] = h.score  # hits arrive best first: the first occurrence is the best score
        out = [ScoredInt(m, sc) for m, sc in best.items()]
        return out if max_matches is None else out[:max_matches]

    def lookup_messages_by_embedding(
        self,
        embedding: NormalizedEmbedding,
        max_matches: int | None = None,
        threshold_score: float | None = None,
        accept_ordinals=None,
    ) -> list[ScoredInt]:
        """`SqliteMessageTextIndex.lookup_by_embedding` / `lookup_in_subset_by_embedding`
        (storage/sqlite/messageindex.py:296-326, 182-257) as ONE device submission: full-corpus top-`max_matches` chunk
        rows (None -> 10, like `fuzzy_lookup_embedding`), THEN the membership filter on their message ordinals
        (`accept_ordinals`, the provider's `ordinals_set`), THEN best score per message, THEN the cut -- the provider's
        order, so the result is the provider's (possibly fewer than `max_matches` messages).  -> [ScoredInt(message, score)]."""
        max_hits, thr = self._limits(max_matches, threshold_score)
        if self._count == 0:
            return []
        eng = self._messages_engine()
        cut = max_hits if max_matches is not None else max(max_hits, 1)
        if eng is None or not (1 <= max_hits <= _PAGE):
            hits = self.fuzzy_lookup_embedding(embedding, max_hits=max_matches, min_score=threshold_score)
            return self._host_rerank(hits, max_matches, None if accept_ordinals is None else set(int(x) for x in accept_ordinals))
        acc = None if accept_ordinals is None else np.fromiter((int(x) for x in accept_ordinals), dtype=np.int64)
        if acc is not None:
            acc = acc[(acc >= 0) & (acc < 2**31 - 1)]
        msgs, scs = eng.search_messages(embedding, max_hits, thr, cut, accept=acc)
        return [ScoredInt(int(m), float(sc)) for m, sc in zip(msgs.tolist(), scs.tolist())]

    def lookup_messages_in_subset_by_embedding(
        self,
        embedding: NormalizedEmbedding,
        rows_of_subset: list[int],
        max_matches: int | None = None,
        threshold_score: float | None = None,
    ) -> list[ScoredInt]:
        """The in-memory provider's form (storage/memory/messageindex.py:173-207 via knowpro/textlocindex.py:164-177): a
        true subset gather over the given index positions, then best score per message, sorted."""
        max_hits, thr = self._limits(max_matches, threshold_score)
        if len(rows_of_subset) == 0 or self._count == 0:
            return []
        eng = self._messages_engine()
        if eng is None or not (1 <= max_hits <= _PAGE):
            hits = self.fuzzy_lookup_embedding_in_subset(embedding, rows_of_subset, max_hits=max_matches, min_score=threshold_score)
            return self._host_rerank(hits, None)
        subset = np.asarray(rows_of_subset)
        if subset.dtype.kind not in "iu":
            raise IndexError("arrays used as indices must be of integer (or boolean) type")
        subset = subset.astype(np.int64, copy=False).reshape(-1)
        n = self._count
        rows = np.where(subset < 0, subset + n, subset)
        bad = (rows < 0) | (rows >= n)
        if bad.any():
            raise IndexError(f"index {int(subset[np.argmax(bad)])} is out of bounds for axis 0 with size {n}")
        msgs, scs = eng.search_messages(embedding, max_hits, thr, max_hits, subset_rows=rows)
        return [ScoredInt(int(m), float(sc)) for m, sc in zip(msgs.tolist(), scs.tolist())]

    async def fuzzy_lookup(
        self,
        key: str,
        max_hits: int | None = None,
        min_score: float | None = None,
        predicate: Callable[[int], bool] | None = None,
    ) -> list[ScoredInt]:
        if max_hits is None:
            max_hits = self.settings.max_matches
        if min_score is None:
            min_score = self.settings.min_score
        embedding = await self.get_embedding(key)
        return self.fuzzy_lookup_embedding(embedding, max_hits=max_hits, min_score=min_score, predicate=predicate)

    # ------------------------------------------------------------------ bookkeeping
    def _set_embedding_size(self, size: int) -> None:
        assert size > 0
        self._embedding_size = size

    def clear(self) -> None:
        self._device_only = None
        self._subset_cache = None
        self._handed_out = False
        self._count = 0
        self._dev_rows = 0
        self._dev_valid = False
        if self._embedding_size > 0:
            self._host = np.zeros((0, self._embedding_size), dtype=np.float32)
        else:
            self._host = np.array([], dtype=np.float32)

    def get_embedding_at(self, pos: int) -> NormalizedEmbedding:
        if 0 <= pos < self._count:
            return self._vectors[pos]
        raise IndexError(f"Index {pos} out of bounds for embedding index of size {len(self)}")

    def serialize_embedding_at(self, pos: int) -> NormalizedEmbedding | None:
        return self._vectors[pos] if 0 <= pos < self._count else None

    def serialize(self) -> NormalizedEmbeddings:
        vectors = self._vectors
        if self._embedding_size > 0:
            assert vectors.shape == (len(vectors), self._embedding_size)
        return vectors  # the live matrix, like the reference (:271)

    def deserialize(self, data: NormalizedEmbeddings | None) -> None:
        if data is None:
            self.clear()
            return
        if self._embedding_size == 0:
            if data.ndim < 2 or data.shape[0] == 0:
                self.clear()  # nothing to learn the width from
                return
            self._set_embedding_size(data.shape[1])
        assert data.shape == (len(data), self._embedding_size), [data.shape, self._embedding_size]
        self._adopt_host(data)  # kept by reference, like the reference (:287)
