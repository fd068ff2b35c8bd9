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
    assert lib.tavb_version() == 1
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
