"""CPU suite: tests/golden/consumer_golden.json (answers of the reference's consumers over the reference class, both verbatim).

  * the comparison rule itself (tests/consumer_scenarios.compare): exact ids, scores within 1e-5, swaps only inside float32 ties;
  * build container only: the file is what the verbatim reference produces TODAY (regenerated and compared), and the verbatim
    consumers over the NEW class (numpy stand-in engine: the class <-> consumer interface, not the kernels) give the same answers;
  * everywhere: `run_replay` -- the product's consumer-side entry points, the code path the GPU test drives -- over the new class on the
    stand-in engine gives the same answers (the GPU twin is tests/test_gpu_consumer_golden.py)."""

import json
import os

import pytest

from oracle import ref_loader, ref_wrappers
from tests import consumer_scenarios as cs
from tests.fake_engine import FakeEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
in_build_container = pytest.mark.skipif(not (ref_loader.reference_available() and ref_wrappers.consumers_available()),
                                        reason="the verbatim reference is only present in the build container")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "consumer_golden.json")) as f:
        return json.load(f)


def test_compare_accepts_ties_only():
    want = {"a": [{"hits": [[3, 0.9], [5, 0.8], [7, 0.8], [9, 0.7]]}], "b": {"hits": [["x", 0.5]]}}
    assert cs.compare(want, want) == (2, 0)
    tie = {"a": [{"hits": [[3, 0.9], [7, 0.8], [5, 0.8], [9, 0.7]]}], "b": {"hits": [["x", 0.5 + 4e-6]]}}
    assert cs.compare(tie, want) == (2, 2)
    bad_cases = {
        "a swap across a real gap": {"a": [{"hits": [[5, 0.9], [3, 0.8], [7, 0.8], [9, 0.7]]}], "b": want["b"]},
        "a result missing": {"a": [{"hits": [[3, 0.9], [5, 0.8], [7, 0.8]]}], "b": want["b"]},
        "a score off by more than 1e-5": {"a": want["a"], "b": {"hits": [["x", 0.5 + 2e-5]]}},
        "a row the reference did not return, without a tie at the cut": {"a": [{"hits": [[3, 0.9], [11, 0.8], [7, 0.8], [9, 0.7]]}], "b": want["b"]},
    }
    for why, bad in bad_cases.items():
        with pytest.raises(AssertionError):
            cs.compare(bad, want)
            pytest.fail(why)
    # a row the reference did not return is legal only in place of a row that ties with the score at the cut (the reference's choice
    # among the rows tied there is numpy's introselect's)
    assert cs.compare({"a": [{"hits": [[3, 0.9], [5, 0.8], [7, 0.8], [11, 0.7]]}], "b": want["b"]}, want) == (2, 1)


@in_build_container
def test_the_file_is_what_the_verbatim_reference_returns(golden):
    ns = ref_wrappers.load_consumers(ref_loader.load_reference_vectorbase())
    assert cs.run_reference(ns) == golden["results"]  # bit for bit: same numpy, same machine class


@in_build_container
def test_verbatim_consumers_over_the_new_class_return_the_file(golden, monkeypatch):
    import typeagent_py_amd.vectorbase as ours
    from typeagent_py_amd import _native

    monkeypatch.setattr(_native, "Engine", FakeEngine)
    lists, swapped = cs.compare(cs.run_reference(ref_wrappers.load_consumers(ours)), golden["results"])
    assert lists >= 200 and swapped <= lists // 20


def test_replay_through_the_adapters_returns_the_file(golden, monkeypatch):
    import typeagent_py_amd.vectorbase as ours
    from typeagent_py_amd import _native, adapters

    monkeypatch.setattr(_native, "Engine", FakeEngine)
    lists, swapped = cs.compare(cs.run_replay(ours, adapters), golden["results"])
    assert lists >= 200 and swapped <= lists // 20
