"""GPU suite: the answers of the reference's VectorBase CONSUMERS -- recorded in the build container from the consumer files executed
verbatim over the verbatim reference class (tests/golden/make_consumer_golden.py -> tests/golden/consumer_golden.json) -- replayed on
the REAL engine through the product's consumer-side entry points: the class `install()` registers as `typeagent.aitools.vectorbase`
and `typeagent_py_amd.adapters` (message aggregation on the device, batched `lookup_terms`, the SQLite BLOB loader).  Nothing here
reads /root/reference.  (The in-container twin, tests/test_consumer_golden.py, runs the verbatim consumers over the new class on the
stand-in engine against the same file.)"""

import hashlib
import json
import os
import sys

import numpy as np
import pytest

from tests import consumer_scenarios as cs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "consumer_golden.json")) as f:
        return json.load(f)


def test_inputs_regenerate_bit_for_bit(golden):
    v, chunks, row_to_msg, queries, _ = cs.message_inputs()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(v) == golden["inputs"]["message_rows_sha256"] and sha(queries) == golden["inputs"]["message_queries_sha256"]
    assert sha(row_to_msg) == golden["inputs"]["row_to_msg_sha256"] and len(chunks) == golden["inputs"]["messages"]
    terms, probes = cs.term_inputs()
    emb = cs.run(cs.create_test_embedding_model(cs.TERM_DIM).get_embeddings(terms[:50] + probes[:10]))
    assert sha(np.asarray(emb, dtype=np.float32)) == golden["inputs"]["term_embeddings_sha256"]


def test_consumer_answers_of_the_verbatim_reference_on_the_real_engine(golden):
    import typeagent_py_amd
    from typeagent_py_amd import _native, adapters

    assert "typeagent" not in sys.modules
    touched = typeagent_py_amd.install()
    try:
        assert touched == ["typeagent.aitools.vectorbase"]
        import typeagent.aitools.vectorbase as installed  # what a typeagent process imports after install()

        got = cs.run_replay(installed, adapters)
    finally:
        typeagent_py_amd.uninstall()
    lists, swapped = cs.compare(got, golden["results"])
    assert lists == json.dumps(golden["results"]).count('"hits"') >= 200
    assert swapped <= lists // 20, f"{swapped} positions differ inside float32 ties: more than ties explain"
    with open("/proc/self/maps") as f:
        assert "libtavb.so" in f.read()  # the HIP library served these lookups
    assert _native.device_count() >= 1
