"""ctypes binding of libtavb.so (C ABI in include/tavb.h) plus the device-side
corpus holder.  PyTorch-ROCm is used only as the owner of device buffers and as
the source of the HIP stream / torch.distributed; every computation on the
lookup path is a hand-written gfx950 kernel reached through this binding.

There is deliberately no CPU fallback here: if libtavb.so is missing or no
MI355X is visible, lookups raise `RuntimeError`.
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

import numpy as np

ABI_VERSION = 6  # include/tavb.h TAVB_ABI_VERSION this binding was written against
TAVB_F32 = 0
TAVB_F16 = 1
MAX_FUSED_K = 256
MAX_STREAM_QUERIES = 8

KERNEL_SCAN, KERNEL_MERGE, KERNEL_MFMA, KERNEL_NORMALIZE, KERNEL_CONVERT, KERNEL_MFMA_SAMPLE, KERNEL_SKINNY, KERNEL_RESCORE, KERNEL_EXCHANGE = range(9)
COMM_ID_BYTES = 128

_LIB_NAME = os.environ.get("TAVB_LIBRARY", "libtavb.so")  # "libtavb_debug.so": the ASan/UBSan host build (`make -C csrc debug`)
_lib = None
_lib_lock = threading.Lock()

# (name, restype, argtypes) for every symbol declared in include/tavb.h
_SIGNATURES = [
    ("tavb_version", c_int, []),
    ("tavb_last_error", c_char_p, []),
    ("tavb_device_count", c_int, [POINTER(c_int)]),
    ("tavb_create", c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    ("tavb_destroy", c_int, [c_void_p]),
    ("tavb_synchronize", c_int, [c_void_p]),
    ("tavb_set_option", c_int, [c_void_p, c_char_p, c_int64]),
    ("tavb_get_option", c_int, [c_void_p, c_char_p, POINTER(c_int64)]),
    ("tavb_set_corpus", c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int64]),
    ("tavb_upload_rows", c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int32]),
    ("tavb_corpus_modified", c_int, [c_void_p, c_int64]),
    ("tavb_normalize_rows_f32", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32]),
    ("tavb_convert_f32_to_f16", c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    ("tavb_search", c_int, [c_void_p, c_void_p, c_int32, c_float, c_void_p, c_void_p, POINTER(c_int32)]),
    ("tavb_search_subset", c_int,
     [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p, POINTER(c_int32)]),
    ("tavb_search_batch", c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("tavb_set_row_messages", c_int, [c_void_p, c_void_p, c_int64, c_int64]),
    ("tavb_search_messages", c_int, [c_void_p, c_void_p, c_int32, c_float, c_void_p, c_int64, c_int32, c_void_p, c_void_p, POINTER(c_int32)]),
    ("tavb_search_messages_subset", c_int,
     [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_int32, c_void_p, c_void_p, POINTER(c_int32)]),
    ("tavb_search_begin", c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    ("tavb_search_end", c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    ("tavb_merge_keys_host", c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    ("tavb_search_all", c_int, [c_void_p, c_void_p, c_float, c_int64, c_void_p, c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    ("tavb_search_subset_all", c_int,
     [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_int64, c_void_p, c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    ("tavb_search_after", c_int,
     [c_void_p, c_void_p, c_int32, c_float, c_float, c_int64, c_void_p, c_void_p, POINTER(c_int32)]),
    ("tavb_search_subset_after", c_int,
     [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_float, c_int64, c_void_p, c_void_p, POINTER(c_int32)]),
    ("tavb_search_device", c_int, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    ("tavb_search_subset_device", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    ("tavb_search_subset_resident", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    ("tavb_merge_device", c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    ("tavb_decode_keys", c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    ("tavb_comm_unique_id", c_int, [c_void_p]),
    ("tavb_comm_init", c_int, [c_void_p, c_void_p, c_int32, c_int32]),
    ("tavb_comm_destroy", c_int, [c_void_p]),
    ("tavb_search_allgather", c_int, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    ("tavb_allgather_merge", c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    ("tavb_remap_key_positions", c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64]),
    ("tavb_profile_enable", c_int, [c_void_p, c_int32]),
    ("tavb_profile_reset", c_int, [c_void_p]),
    ("tavb_profile_read", c_int, [c_void_p, c_int32, POINTER(c_double), POINTER(c_int64)]),
    ("tavb_plan_ladder", c_int, [c_int64, c_int32, c_int32, POINTER(c_int64), c_int32]),
]

ABI_SYMBOLS = [name for name, _, _ in _SIGNATURES]


_AS_POINTER = ctypes.c_char * 0  # `_AS_POINTER.from_buffer(ndarray)`: a ctypes object at the array's address, passed where a pointer is expected


def _addr(arr: np.ndarray):
    """The array's address as a ctypes argument: 0.25 us through the buffer protocol (writable arrays) against ~2 us for `ndarray.ctypes.data_as()`."""
    try:
        return _AS_POINTER.from_buffer(arr)
    except (TypeError, ValueError, BufferError):  # read-only (or otherwise not exportable) buffer
        return arr.ctypes.data_as(c_void_p)


def plan_ladder(rows: int, nq: int, n_cu: int = 256) -> list[int]:
    """Phase boundaries of the threshold ladder the library runs for a batch of `nq` (>= 65; >= 33 on corpora of 256 MiB and more) queries over `rows` rows with default options
    (tavb_plan_ladder): len(result) - 1 = tile-kernel launches per lookup.  Needs no GPU."""
    lib = load_library(preload_torch=False)
    buf = (c_int64 * 16)()
    n = lib.tavb_plan_ladder(int(rows), int(nq), int(n_cu), buf, 16)
    _check(lib, n if n < 0 else 0)
    return [int(buf[i]) for i in range(n + 1)]


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load_library(preload_torch: bool = True):
    """dlopen libtavb.so and declare its prototypes.

    torch is imported first (when available) so that the process holds ONE HIP
    runtime: torch's bundled libamdhip64.so has the same SONAME
    (libamdhip64.so.7) as the ROCm one libtavb.so was linked against, so the
    dynamic loader reuses the copy that is already mapped.
    """
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.isfile(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C typeagent_py_amd/csrc`). There is no CPU fallback."
            )
        if preload_torch:
            try:
                import torch  # noqa: F401
            except Exception:  # pragma: no cover - torch is part of the image
                pass
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        try:
            lib.tavb_version.restype = c_int
            lib.tavb_version.argtypes = []
            found = int(lib.tavb_version())
        except AttributeError as exc:
            raise RuntimeError(f"{path} is not a libtavb build (no tavb_version): rebuild it with `make -C typeagent_py_amd/csrc`") from exc
        if found != ABI_VERSION:
            raise RuntimeError(f"{path} implements C ABI version {found}, this binding needs {ABI_VERSION} (include/tavb.h): "
                               "a stale build -- run `make -C typeagent_py_amd/csrc` (or check TAVB_LIBRARY)")
        for name, restype, argtypes in _SIGNATURES:
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return lib


class TavbError(RuntimeError):
    """A libtavb call failed (HIP/RCCL failures surface here, never as abort())."""


class TavbTimeout(TavbError):
    """`synchronize()`: an exchange of a collective lookup did not complete within the `comm_timeout_ms` option -- a peer never joined.  The
    library has aborted its communicator; `comm_init` again to rejoin."""


def _check(lib, rc: int) -> None:
    if rc != 0:
        msg = lib.tavb_last_error()
        text = msg.decode("utf-8", "replace") if msg else "unknown error"
        if rc == -1:
            raise ValueError(f"libtavb: {text}")
        if rc == -7:
            raise TavbTimeout(f"libtavb error {rc}: {text}")
        raise TavbError(f"libtavb error {rc}: {text}")


def device_count() -> int:
    lib = load_library()
    n = c_int(0)
    rc = lib.tavb_device_count(byref(n))
    if rc != 0:
        return 0
    return int(n.value)


def f32_threshold(min_score) -> np.float32:
    """The float32 `t` with (s >= t) == numpy's (s >= min_score) for float32 `s`.

    vectorbase.py:179 compares a float32 array with the caller's scalar.  Under
    NEP 50 a Python float/int is a weak scalar and is cast to float32 first
    (0.85 -> 0.8500000238...); a numpy float64 scalar forces a float64 compare,
    which equals comparing with the smallest float32 that is >= it.
    """
    if type(min_score) is float and -3.0e38 < min_score < 3.0e38:  # the common case, without the errstate context (1 us of a 30 us lookup)
        return np.float32(min_score)
    if isinstance(min_score, np.floating) and not isinstance(min_score, np.float32):
        wide = float(min_score)
        if wide != wide:
            return np.float32(np.nan)
        narrow = np.float32(wide)
        if float(narrow) < wide:
            narrow = np.nextafter(narrow, np.float32(np.inf), dtype=np.float32)
        return np.float32(narrow)
    with np.errstate(over="ignore"):
        return np.float32(min_score)


class Engine:
    """One libtavb context on one GPU + the torch tensors that hold the corpus."""

    def __init__(self, device: int | None = None, use_torch_stream: bool = False):
        import torch

        self._torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError(
                "typeagent_py_amd: no HIP device visible (torch.cuda.is_available() is False); "
                "the VectorBase engine runs on MI355X only and has no CPU fallback"
            )
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        stream_ptr = None
        if use_torch_stream:
            stream_ptr = c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        handle = c_void_p()
        _check(self.lib, self.lib.tavb_create(self.device, stream_ptr, byref(handle)))
        self._h = handle
        self._lock = threading.Lock()
        self._out_cache: dict = {}  # k -> reused output arrays of the single-query calls (+ their ctypes views)
        self.corpus = None  # torch tensor [capacity, dim]
        self._owns_corpus = False  # True when upload_rows allocated it (a tensor adopted from the caller is never appended into)
        self.rows = 0
        self.dim = 0
        self.dtype = TAVB_F32
        self.ordinal_base = 0
        # measurement / debugging hook: TAVB_ENGINE_OPTIONS="name=value,name=value" is applied to every context of the process (e.g. the GPU
        # suite under another kernel variant: TAVB_ENGINE_OPTIONS=mfma_sched=8 pytest -m gpu).  An unknown name raises, as set_option does.
        for item in filter(None, (x.strip() for x in os.environ.get("TAVB_ENGINE_OPTIONS", "").split(","))):
            name, _, val = item.partition("=")
            self.set_option(name.strip(), int(val))

    # -- lifecycle ---------------------------------------------------------
    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h is not None and self.lib is not None:
            self.lib.tavb_destroy(h)
        self.corpus = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- options / profiling ----------------------------------------------
    def set_option(self, name: str, value: int) -> None:
        _check(self.lib, self.lib.tavb_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        out = c_int64(0)
        _check(self.lib, self.lib.tavb_get_option(self._h, name.encode(), byref(out)))
        return int(out.value)

    def profile_enable(self, on: bool = True) -> None:
        _check(self.lib, self.lib.tavb_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self) -> None:
        _check(self.lib, self.lib.tavb_profile_reset(self._h))

    def profile_read(self, kernel_id: int) -> tuple[float, int]:
        ms, n = c_double(0.0), c_int64(0)
        _check(self.lib, self.lib.tavb_profile_read(self._h, kernel_id, byref(ms), byref(n)))
        return float(ms.value), int(n.value)

    def synchronize(self) -> None:
        _check(self.lib, self.lib.tavb_synchronize(self._h))

    # -- corpus ------------------------------------------------------------
    def torch_dtype(self, dtype: int):
        return self._torch.float16 if dtype == TAVB_F16 else self._torch.float32

    def set_corpus_tensor(self, tensor, rows: int | None = None, ordinal_base: int = 0, sync_torch: bool = True, _owned: bool = False) -> None:
        """Adopt a device tensor [>=rows, dim] (f32 or f16, contiguous) as the corpus."""
        torch = self._torch
        if tensor.dim() != 2 or not tensor.is_contiguous():
            raise ValueError("corpus tensor must be a contiguous 2-D tensor")
        if tensor.device.type != "cuda" or (tensor.device.index or 0) != self.device:
            raise ValueError(f"corpus tensor must live on cuda:{self.device}")
        if tensor.dtype == torch.float32:
            dt = TAVB_F32
        elif tensor.dtype == torch.float16:
            dt = TAVB_F16
        else:
            raise ValueError("corpus tensor must be float32 or float16")
        n = tensor.shape[0] if rows is None else int(rows)
        if n > tensor.shape[0]:
            raise ValueError("rows exceeds the tensor")
        # make sure whatever produced the tensor on torch's stream has finished
        if sync_torch:
            torch.cuda.current_stream(self.device).synchronize()
        _check(self.lib, self.lib.tavb_set_corpus(self._h, c_void_p(tensor.data_ptr()), n, tensor.shape[1], dt, int(ordinal_base)))
        if not _owned and tensor is not self.corpus:
            # a caller's tensor: the allocator may have put it where a freed one of the same shape lived, and the library keys its
            # per-corpus caches (row-norm maxima, the fp16 shadow of an fp32 corpus) on the address
            _check(self.lib, self.lib.tavb_corpus_modified(self._h, 0))
        self.corpus, self.rows, self.dim, self.dtype, self.ordinal_base = tensor, n, int(tensor.shape[1]), dt, int(ordinal_base)
        self._owns_corpus = _owned

    def upload_rows(self, host_rows: np.ndarray, start: int, dtype: int, capacity_hint: int = 0) -> None:
        """Make device rows [start, start+len) equal to `host_rows` (f32), growing the
        buffer geometrically; rows [0, start) are kept (vectorbase.py:128/145 re-copies
        the whole matrix on every add; here an append moves only the new rows)."""
        torch = self._torch
        n_new = start + host_rows.shape[0]
        dim = host_rows.shape[1]
        tdt = self.torch_dtype(dtype)
        dev = torch.device("cuda", self.device)
        need_new = (
            self.corpus is None or self.corpus.shape[1] != dim or self.corpus.dtype != tdt or self.corpus.shape[0] < n_new
            or not self._owns_corpus  # adopted from the caller: copy into a buffer of our own before writing rows
        )
        if need_new:
            cap = max(n_new, capacity_hint, 2 * (self.corpus.shape[0] if self.corpus is not None and start > 0 else 0), 16)
            fresh = torch.empty((cap, dim), dtype=tdt, device=dev)
            if start > 0:
                if self.corpus is None or self.corpus.shape[1] != dim or self.corpus.dtype != tdt:
                    raise RuntimeError("cannot keep old rows across a layout change")
                fresh[:start].copy_(self.corpus[:start])
            self.corpus = fresh
        if host_rows.shape[0]:
            # pinned double-buffered staging + async H2D + on-device f32 -> f16 (tavb_upload_rows): no fp16 / second fp32 host copy
            src = np.ascontiguousarray(host_rows, dtype=np.float32)
            torch.cuda.current_stream(self.device).synchronize()  # the allocation / copy of old rows above ran on torch's stream
            dst = self.corpus.data_ptr() + start * dim * (2 if dtype == TAVB_F16 else 4)
            _check(self.lib, self.lib.tavb_upload_rows(self._h, _addr(src), src.shape[0], dim, c_void_p(dst), dtype))
        self.set_corpus_tensor(self.corpus, rows=n_new, ordinal_base=self.ordinal_base, _owned=True)
        _check(self.lib, self.lib.tavb_corpus_modified(self._h, int(start)))  # rows [start, n_new) were (re)written
        return True

    def clear(self) -> None:
        if self.corpus is not None:
            self.set_corpus_tensor(self.corpus, rows=0, ordinal_base=self.ordinal_base, _owned=self._owns_corpus)
        self.rows = 0

    # -- K1 / convert --------------------------------------------------------
    def normalize_rows_(self, tensor) -> None:
        torch = self._torch
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.dim() == 2
        torch.cuda.current_stream(self.device).synchronize()
        _check(self.lib, self.lib.tavb_normalize_rows_f32(self._h, c_void_p(tensor.data_ptr()), c_void_p(tensor.data_ptr()),
                                                          tensor.shape[0], tensor.shape[1]))
        self.synchronize()

    def normalize_rows(self, tensor):
        out = self._torch.empty_like(tensor)
        self._torch.cuda.current_stream(self.device).synchronize()
        _check(self.lib, self.lib.tavb_normalize_rows_f32(self._h, c_void_p(tensor.data_ptr()), c_void_p(out.data_ptr()),
                                                          tensor.shape[0], tensor.shape[1]))
        self.synchronize()
        return out

    def to_f16(self, tensor):
        torch = self._torch
        assert tensor.dtype == torch.float32 and tensor.is_contiguous()
        out = torch.empty(tensor.shape, dtype=torch.float16, device=tensor.device)
        torch.cuda.current_stream(self.device).synchronize()
        _check(self.lib, self.lib.tavb_convert_f32_to_f16(self._h, c_void_p(tensor.data_ptr()), c_void_p(out.data_ptr()), tensor.numel()))
        self.synchronize()
        return out

    # -- lookups -------------------------------------------------------------
    def _query(self, q) -> np.ndarray:
        a = np.ascontiguousarray(q, dtype=np.float32)
        if a.ndim != 1 or a.shape[0] != self.dim:
            raise ValueError(f"shapes ({self.rows},{self.dim}) and {tuple(np.shape(q))} not aligned: query must have {self.dim} elements")
        return a

    def _out_buffers(self, k: int):
        """Reused output arrays of the single-query calls, with their ctypes views: `ndarray.ctypes.data_as()` costs ~2 us a piece on the host --
        three of them were 6 us of a 33 us lookup on a 10k-row corpus.  Callers hold self._lock and copy the filled prefix out."""
        buf = self._out_cache.get(k)
        if buf is None:
            ords = np.empty(k, dtype=np.int64)
            scs = np.empty(k, dtype=np.float32)
            buf = self._out_cache[k] = (ords, scs, _AS_POINTER.from_buffer(ords), _AS_POINTER.from_buffer(scs), c_int32(0))
            if len(self._out_cache) > 16:
                self._out_cache.pop(next(iter(self._out_cache)))
        return buf

    def search(self, q, k: int, thr: np.float32, after: tuple[float, int] | None = None):
        """-> (ordinals int64[m], scores float32[m]) best first, m <= k <= MAX_FUSED_K."""
        a = self._query(q)
        try:
            pa = _AS_POINTER.from_buffer(a)  # 0.25 us; needs a writable buffer
        except (TypeError, ValueError):
            pa = _addr(a)
        with self._lock:
            ords, scs, p_ords, p_scs, cnt = self._out_buffers(k)
            if after is None:
                rc = self.lib.tavb_search(self._h, pa, k, c_float(float(thr)), p_ords, p_scs, byref(cnt))
            else:
                rc = self.lib.tavb_search_after(self._h, pa, k, c_float(float(thr)), c_float(after[0]), int(after[1]), p_ords, p_scs, byref(cnt))
            if rc == 0:
                m = int(cnt.value)
                out = ords[:m].copy(), scs[:m].copy()
        _check(self.lib, rc)
        return out

    def search_subset(self, q, rows: np.ndarray, k: int, thr: np.float32, after: tuple[float, int] | None = None):
        """rows: int64 corpus rows per subset position -> (positions int64[m], scores float32[m])."""
        a = self._query(q)
        r = np.ascontiguousarray(rows, dtype=np.int64)
        pos = np.empty(k, dtype=np.int64)
        scs = np.empty(k, dtype=np.float32)
        cnt = c_int32(0)
        with self._lock:
            if after is None:
                rc = self.lib.tavb_search_subset(self._h, _addr(a), _addr(r), r.shape[0], k,
                                                 c_float(float(thr)), _addr(pos), _addr(scs), byref(cnt))
            else:
                rc = self.lib.tavb_search_subset_after(self._h, _addr(a), _addr(r), r.shape[0], k,
                                                       c_float(float(thr)), c_float(after[0]), int(after[1]),
                                                       _addr(pos), _addr(scs), byref(cnt))
        _check(self.lib, rc)
        m = int(cnt.value)
        return pos[:m], scs[:m]

    def rows_to_device(self, rows: np.ndarray):
        """int64 corpus rows (already wrapped and range-checked) -> torch int32 [S] on this engine's device."""
        torch = self._torch
        return torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(torch.device("cuda", self.device))

    def search_subset_resident(self, q, dev_rows, k: int, thr: np.float32):
        """`search_subset` over a row list that is already on the device (torch int32 [S], wrapped and range-checked by the caller)."""
        a = self._query(q)
        pos = np.empty(k, dtype=np.int64)
        scs = np.empty(k, dtype=np.float32)
        cnt = c_int32(0)
        with self._lock:
            rc = self.lib.tavb_search_subset_resident(self._h, _addr(a), c_void_p(dev_rows.data_ptr()), int(dev_rows.shape[0]), k,
                                                      c_float(float(thr)), _addr(pos), _addr(scs), byref(cnt))
        _check(self.lib, rc)
        m = int(cnt.value)
        return pos[:m], scs[:m]

    def search_all(self, q, thr: np.float32, max_out: int | None = None, subset_rows=None):
        """Every row (or subset position) with score >= thr, best first, in one corpus pass; the first `max_out` of them
        (None: all) -> (ordinals-or-positions int64[m], scores float32[m])."""
        a = self._query(q)
        r = None if subset_rows is None else np.ascontiguousarray(subset_rows, dtype=np.int64)
        cap = (self.rows if r is None else r.shape[0]) if max_out is None else min(int(max_out), self.rows if r is None else r.shape[0])
        items = np.empty(max(cap, 1), dtype=np.int64)
        scs = np.empty(max(cap, 1), dtype=np.float32)
        cnt, total = c_int64(0), c_int64(0)
        with self._lock:
            if r is None:
                rc = self.lib.tavb_search_all(self._h, _addr(a), c_float(float(thr)), cap, _addr(items),
                                              _addr(scs), byref(cnt), byref(total))
            else:
                rc = self.lib.tavb_search_subset_all(self._h, _addr(a), _addr(r), r.shape[0], c_float(float(thr)), cap,
                                                     _addr(items), _addr(scs), byref(cnt), byref(total))
        _check(self.lib, rc)
        m = int(cnt.value)
        return items[:m], scs[:m]

    def search_batch(self, queries, k: int, thrs):
        """queries f32 [nq, dim]; thrs float32 [nq] -> (ordinals [nq,k], scores [nq,k], counts [nq])."""
        a = np.ascontiguousarray(queries, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] != self.dim:
            raise ValueError(f"queries must be [nq, {self.dim}]")
        nq = a.shape[0]
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(thrs, dtype=np.float32), (nq,)))
        ords = np.empty((nq, k), dtype=np.int64)
        scs = np.empty((nq, k), dtype=np.float32)
        cnts = np.zeros(nq, dtype=np.int32)
        with self._lock:  # (five `ndarray.ctypes.data_as()` were 10 us of a 32 us two-query lookup on a small corpus)
            rc = self.lib.tavb_search_batch(self._h, _addr(a), nq, k, _addr(t), _addr(ords), _addr(scs), _addr(cnts))
        _check(self.lib, rc)
        return ords, scs, cnts

    # message re-rank on the device ---------------------------------------------
    def set_row_messages(self, row_to_message: np.ndarray) -> None:
        """int array [rows]: chunk row -> message ordinal (-1 = none); kept on the device next to the corpus."""
        torch = self._torch
        m = np.ascontiguousarray(row_to_message, dtype=np.int32)
        if m.ndim != 1:
            raise ValueError("row_to_message must be 1-D")
        self.row_messages = torch.from_numpy(m).to(torch.device("cuda", self.device))
        torch.cuda.current_stream(self.device).synchronize()
        n_messages = int(m.max()) + 1 if m.size else 0
        _check(self.lib, self.lib.tavb_set_row_messages(self._h, c_void_p(self.row_messages.data_ptr()), m.shape[0], max(n_messages, 0)))

    def search_messages(self, q, k: int, thr: np.float32, max_messages: int, accept=None, subset_rows=None):
        """-> (message ordinals int64[m], scores float32[m]).  accept: int array of accepted message ordinals (sqlite
        provider's subset form) or None; subset_rows: int64 corpus rows (memory provider's subset gather) or None."""
        a = self._query(q)
        msgs = np.empty(k, dtype=np.int64)
        scs = np.empty(k, dtype=np.float32)
        cnt = c_int32(0)
        with self._lock:
            if subset_rows is not None:
                r = np.ascontiguousarray(subset_rows, dtype=np.int64)
                rc = self.lib.tavb_search_messages_subset(self._h, _addr(a), _addr(r), r.shape[0], k, c_float(float(thr)),
                                                          int(max_messages), _addr(msgs), _addr(scs), byref(cnt))
            else:
                acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.int32)
                rc = self.lib.tavb_search_messages(self._h, _addr(a), k, c_float(float(thr)),
                                                   _addr(acc) if acc is not None and acc.size else None,
                                                   -1 if acc is None else acc.shape[0], int(max_messages),
                                                   _addr(msgs), _addr(scs), byref(cnt))
        _check(self.lib, rc)
        m = int(cnt.value)
        return msgs[:m], scs[:m]

    # split form (several contexts driven from one thread) --------------------
    def search_begin(self, queries: np.ndarray, k: int, thrs: np.ndarray, cursor_key: int | None = None) -> None:
        """queries f32 [nq, dim] (contiguous), thrs f32 [nq]: enqueue; pair with search_end(nq, k)."""
        cur = None
        if cursor_key is not None:
            cur = (c_uint64 * 1)(int(cursor_key))
        with self._lock:
            rc = self.lib.tavb_search_begin(self._h, _addr(queries), queries.shape[0], k, _addr(thrs),
                                            ctypes.cast(cur, c_void_p) if cur is not None else None)
        _check(self.lib, rc)

    def search_end(self, nq: int, k: int, out_keys: np.ndarray) -> None:
        """out_keys: uint64 [nq, k] (contiguous) <- sorted key lists with global ordinals."""
        with self._lock:
            rc = self.lib.tavb_search_end(self._h, nq, k, _addr(out_keys))
        _check(self.lib, rc)

    # device-resident forms ---------------------------------------------------
    def search_device(self, dev_queries, k: int, thr: float, out_keys=None):
        """dev_queries: torch f32 [nq, dim] on this device -> torch int64 [nq, k] of packed keys (async)."""
        torch = self._torch
        assert dev_queries.dtype == torch.float32 and dev_queries.is_contiguous() and dev_queries.shape[1] == self.dim
        nq = dev_queries.shape[0]
        if out_keys is None:
            out_keys = torch.empty((nq, k), dtype=torch.int64, device=dev_queries.device)
        with self._lock:
            rc = self.lib.tavb_search_device(self._h, c_void_p(dev_queries.data_ptr()), nq, k, c_float(float(thr)), c_void_p(out_keys.data_ptr()))
        _check(self.lib, rc)
        return out_keys

    def search_subset_device(self, dev_query, dev_rows, k: int, thr: float, out_keys=None):
        """dev_query: torch f32 [1, dim] or [dim]; dev_rows: torch int32 [S] (valid corpus rows) ->
        torch int64 [1, k] packed keys carrying subset positions (async)."""
        torch = self._torch
        assert dev_query.dtype == torch.float32 and dev_query.is_contiguous() and dev_query.numel() == self.dim
        assert dev_rows.dtype == torch.int32 and dev_rows.is_contiguous() and dev_rows.dim() == 1
        if out_keys is None:
            out_keys = torch.empty((1, k), dtype=torch.int64, device=dev_query.device)
        with self._lock:
            rc = self.lib.tavb_search_subset_device(self._h, c_void_p(dev_query.data_ptr()), c_void_p(dev_rows.data_ptr()),
                                                    dev_rows.shape[0], k, c_float(float(thr)), c_void_p(out_keys.data_ptr()))
        _check(self.lib, rc)
        return out_keys

    # row shards, one process per GPU: the library's own RCCL communicator ----------------------------------
    def comm_init(self, unique_id: bytes, rank: int, world: int) -> None:
        """Collective over all ranks: join the communicator named by `unique_id` (from `comm_unique_id()` on rank 0)."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"the rendezvous id is {COMM_ID_BYTES} bytes")
        buf = ctypes.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        with self._lock:
            rc = self.lib.tavb_comm_init(self._h, ctypes.cast(buf, c_void_p), int(rank), int(world))
        _check(self.lib, rc)

    def comm_destroy(self) -> None:
        with self._lock:
            rc = self.lib.tavb_comm_destroy(self._h)
        _check(self.lib, rc)

    def search_allgather(self, dev_queries, k: int, thr: float, out_keys=None):
        """Collective `search_device`: every rank passes the same queries and gets the merged whole-corpus key lists [nq, k].
        `out_keys`: a device tensor or a PINNED host tensor (int64 [nq, k]; the merge kernel writes it directly).  Async."""
        torch = self._torch
        assert dev_queries.dtype == torch.float32 and dev_queries.is_contiguous() and dev_queries.shape[1] == self.dim
        nq = dev_queries.shape[0]
        if out_keys is None:
            out_keys = torch.empty((nq, k), dtype=torch.int64, device=dev_queries.device)
        with self._lock:
            rc = self.lib.tavb_search_allgather(self._h, c_void_p(dev_queries.data_ptr()), nq, k, c_float(float(thr)), c_void_p(out_keys.data_ptr()))
        _check(self.lib, rc)
        return out_keys

    def allgather_merge(self, dev_local_keys, out_keys=None):
        """Collective: this rank's sorted lists [nq, k] (global ordinals / positions) -> the lists merged over all ranks (async)."""
        torch = self._torch
        assert dev_local_keys.dtype == torch.int64 and dev_local_keys.is_contiguous() and dev_local_keys.dim() == 2
        nq, k = dev_local_keys.shape
        if out_keys is None:
            out_keys = torch.empty((nq, k), dtype=torch.int64, device=dev_local_keys.device)
        with self._lock:
            rc = self.lib.tavb_allgather_merge(self._h, c_void_p(dev_local_keys.data_ptr()), nq, k, c_void_p(out_keys.data_ptr()))
        _check(self.lib, rc)
        return out_keys

    def remap_key_positions(self, dev_keys, dev_map) -> None:
        """In place: keys carrying list positions -> keys carrying dev_map[position] (int32 device tensor).  Async."""
        torch = self._torch
        assert dev_keys.dtype == torch.int64 and dev_keys.is_contiguous() and dev_map.dtype == torch.int32 and dev_map.is_contiguous()
        with self._lock:
            rc = self.lib.tavb_remap_key_positions(self._h, c_void_p(dev_keys.data_ptr()), dev_keys.numel(), c_void_p(dev_map.data_ptr()), dev_map.numel())
        _check(self.lib, rc)

    def merge_device(self, dev_lists, out_keys=None):
        """dev_lists: torch int64 [n_lists, nq, k] -> [nq, k] (async)."""
        torch = self._torch
        assert dev_lists.dtype == torch.int64 and dev_lists.is_contiguous() and dev_lists.dim() == 3
        n_lists, nq, k = dev_lists.shape
        if out_keys is None:
            out_keys = torch.empty((nq, k), dtype=torch.int64, device=dev_lists.device)
        with self._lock:
            rc = self.lib.tavb_merge_device(self._h, c_void_p(dev_lists.data_ptr()), n_lists, nq, k, c_void_p(out_keys.data_ptr()))
        _check(self.lib, rc)
        return out_keys


def comm_unique_id() -> bytes:
    """A fresh RCCL rendezvous id (rank 0 creates it and hands it to the other ranks)."""
    lib = load_library()
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(lib, lib.tavb_comm_unique_id(ctypes.cast(buf, c_void_p)))
    return buf.raw


def make_key(score: float, ordinal: int) -> int:
    """The packed key of (score, ordinal): (float32 bits << 32) | (0xFFFFFFFF - ordinal)."""
    bits = int(np.float32(score).view(np.uint32))
    return (bits << 32) | (0xFFFFFFFF - int(ordinal))


def merge_keys(lists: np.ndarray) -> np.ndarray:
    """uint64 [n_lists, nq, k] sorted key lists -> uint64 [nq, k] (host k-way merge in libtavb)."""
    lib = load_library(preload_torch=False)
    a = np.ascontiguousarray(lists, dtype=np.uint64)
    n_lists, nq, k = a.shape
    out = np.empty((nq, k), dtype=np.uint64)
    _check(lib, lib.tavb_merge_keys_host(_addr(a), n_lists, nq, k, _addr(out)))
    return out


def decode_keys(keys: np.ndarray):
    """Host int64/uint64 [nq, k] packed keys -> (ordinals [nq,k], scores [nq,k], counts [nq])."""
    lib = load_library(preload_torch=False)
    a = np.ascontiguousarray(keys).view(np.uint64)
    if a.ndim != 2:
        raise ValueError("keys must be 2-D")
    nq, k = a.shape
    ords = np.empty((nq, k), dtype=np.int64)
    scs = np.empty((nq, k), dtype=np.float32)
    cnts = np.zeros(nq, dtype=np.int32)
    _check(lib, lib.tavb_decode_keys(_addr(a), nq, k, _addr(ords),
                                     _addr(scs), _addr(cnts)))
    return ords, scs, cnts
