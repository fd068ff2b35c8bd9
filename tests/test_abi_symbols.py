"""CPU suite: libtavb.so loads and exports every function include/tavb.h declares
(no compute calls: there is no GPU here)."""

import ctypes
import os
import re

import pytest

from typeagent_py_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions() -> list[str]:
    text = open(os.path.join(ROOT, "include", "tavb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tavb_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_native.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    path = _native.library_path()
    assert os.path.isfile(path), "libtavb.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} missing from libtavb.so"


def test_version_and_error_string_and_host_helpers():
    lib = _native.load_library()
    header = open(os.path.join(ROOT, "include", "tavb.h")).read()
    assert lib.tavb_version() == _native.ABI_VERSION == int(re.search(r"#define TAVB_ABI_VERSION (\d+)", header).group(1))
    # a failing call sets a readable message and never aborts
    rc = lib.tavb_destroy(None)
    assert rc == 0
    rc = lib.tavb_set_option(None, b"scan_waves", 4)
    assert rc == -1 and b"null context" in lib.tavb_last_error()
    # pure host helper: decode packed keys
    import numpy as np

    def key(score, idx):
        return (int(np.float32(score).view(np.uint32)) << 32) | (0xFFFFFFFF - idx)

    keys = np.array([[key(1.0, 7), key(0.5, 3), 0, 0], [0, 0, 0, 0]], dtype=np.uint64)
    ords, scs, cnts = _native.decode_keys(keys)
    assert cnts.tolist() == [2, 0]
    assert ords[0, :2].tolist() == [7, 3] and scs[0, :2].tolist() == [1.0, 0.5]


def test_engine_refuses_to_start_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        _native.Engine()


def test_a_library_of_another_abi_version_is_refused(tmp_path, monkeypatch):
    """Round-2 advice: a stale libtavb.so (or one picked through TAVB_LIBRARY) used to fail with an AttributeError at bind time or
    mis-index the profile slots; now the version is compared first, with a message that says what to do."""
    import subprocess

    src = tmp_path / "stale.c"
    src.write_text("int tavb_version(void) { return 1; }\n")
    so = tmp_path / "libtavb_stale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "library_path", lambda: str(so))
    with pytest.raises(RuntimeError, match="C ABI version 1, this binding needs"):
        _native.load_library()


def test_plan_ladder_is_a_partition_of_the_rows():
    """tavb_plan_ladder (no GPU needed): the threshold ladder's phase boundaries start at 0, end at the row count, grow strictly, and a corpus
    too small for a seeding phase (below 30720 rows) is scanned in one phase; bad shapes come back as errors."""
    import pytest

    from typeagent_py_amd import _native

    for rows in (1, 319, 10_000, 30_719, 30_720, 50_000, 81_919, 81_920, 163_840, 1_000_000, 1_250_000, 10_000_000, 100_000_000):
        for nq in (65, 128, 256, 300, 1024, 4096):
            b = _native.plan_ladder(rows, nq)
            assert b[0] == 0 and b[-1] == rows and all(x < y for x, y in zip(b, b[1:])), (rows, nq, b)
            assert 1 <= len(b) - 1 <= 8
            if len(b) > 2:
                assert rows >= 8 * b[1]            # a seeding phase only when the corpus is at least 8 samples long
                assert b[-1] - b[-2] >= b[-2]      # the last phase is at least as long as everything before it
    assert len(_native.plan_ladder(20_000, 1024)) == 2                     # one phase
    assert _native.plan_ladder(50_000, 1024) == [0, 6080, 50_000]          # a small corpus: an eighth of the rows in whole tiles, then the rest (round 6)
    assert _native.plan_ladder(150_000, 128) == [0, 18560, 150_000]
    assert _native.plan_ladder(200_000, 128) == [0, 20480, 200_000]  # eight first phases' worth and more: the generic ladder
    assert _native.plan_ladder(10_000_000, 1024) == [0, 10240, 51200, 256000, 1280000, 10_000_000]
    with pytest.raises(ValueError):
        _native.plan_ladder(-1, 1024)
    with pytest.raises(ValueError):
        _native.plan_ladder(1000, 0)


def test_array_arguments_go_through_the_buffer_protocol_or_fall_back():
    """`_native._addr`: the address ctypes receives is the array's own, for writable arrays (buffer protocol), read-only ones, empty ones and
    views that cannot export a contiguous buffer (the `ndarray.ctypes` fallback)."""
    import ctypes

    import numpy as np

    from typeagent_py_amd import _native

    def address(arg):
        return ctypes.cast(arg, ctypes.c_void_p).value if not isinstance(arg, ctypes.c_void_p) else arg.value

    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert address(_native._addr(a)) == a.ctypes.data
    ro = a.copy()
    ro.setflags(write=False)
    assert address(_native._addr(ro)) == ro.ctypes.data
    assert address(_native._addr(a[1:])) == a[1:].ctypes.data  # a contiguous slice: its own start
    strided = a[:, ::2]
    assert address(_native._addr(strided)) == strided.ctypes.data
    empty = np.empty(0, dtype=np.int64)
    _native._addr(empty)  # (nothing to read or write: any pointer does; must not raise)
    # and the library reads / writes through it
    lib = _native.load_library(preload_torch=False)
    lists = np.array([[[9, 4, 1]], [[7, 6, 0]]], dtype=np.uint64)
    out = np.zeros((1, 3), dtype=np.uint64)
    assert lib.tavb_merge_keys_host(_native._addr(lists), 2, 1, 3, _native._addr(out)) == 0
    assert out.tolist() == [[9, 7, 6]]


def test_every_workspace_of_a_context_is_released_by_tavb_destroy():
    """`Buffer` members of the context are freed one by one in tavb_destroy (no destructor: the device must be current): round 5 found the
    fp16 shadow -- half an fp32 corpus' bytes -- missing from that list.  Source-level check, so that the next buffer cannot be forgotten."""
    import re

    src = open(os.path.join(ROOT, "typeagent_py_amd", "csrc", "tavb_abi.hip")).read()
    names = set()
    for m in re.finditer(r"^\s*Buffer\s+([^;]+);", src, re.M):
        for part in m.group(1).split(","):
            mm = re.match(r"\s*([dh]_[a-z_0-9]*)", part)
            if mm:
                names.add(mm.group(1))
    assert {"d_shadow", "d_queries_pad", "h_ring", "d_gather"} <= names and len(names) >= 25
    start = src.index("int tavb_destroy(tavb_ctx* c) {")
    body = src[start : src.index("\n}\n", start)]
    released = set(re.findall(r"c->([a-z_0-9]+)(?:\[i\])?\.release\(\)", body))
    assert names <= released, sorted(names - released)
