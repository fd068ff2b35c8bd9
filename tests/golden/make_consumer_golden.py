#!/usr/bin/env python3
"""Generate tests/golden/consumer_golden.json: what the reference's VectorBase CONSUMERS return, executed VERBATIM
(oracle/ref_wrappers.py) over the VERBATIM reference `VectorBase` (oracle/ref_loader.py) for the scenarios of
tests/consumer_scenarios.py.  Build container only (needs /root/reference):

    python tests/golden/make_consumer_golden.py

Neither our oracle nor our engine takes part.  The inputs are not stored: they are regenerated from seeds (tests/synth.py) and
by the hash embedder of tests/fakes.py, and pinned here by sha256 so that a drifting generator is detected, not compared.
"""

from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_loader, ref_wrappers  # noqa: E402
from tests import consumer_scenarios as cs  # noqa: E402


def input_digest() -> dict:
    v, chunks, row_to_msg, queries, subsets = cs.message_inputs()
    terms, probes = cs.term_inputs()
    emb = cs.run(cs.create_test_embedding_model(cs.TERM_DIM).get_embeddings(terms[:50] + probes[:10]))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    return {"message_rows_sha256": sha(v), "message_queries_sha256": sha(queries), "row_to_msg_sha256": sha(row_to_msg), "messages": len(chunks),
            "term_embeddings_sha256": sha(np.asarray(emb, dtype=np.float32))}


def main() -> None:
    ns = ref_wrappers.load_consumers(ref_loader.load_reference_vectorbase())
    out = {
        "generated_by": "tests/golden/make_consumer_golden.py: reference consumers (verbatim) over the reference VectorBase (verbatim)",
        "reference_files": sorted(list(ref_wrappers.CONSUMER_FILES.values()) + list(ref_wrappers.TRANSFORMED_FILES.values())),
        "inputs": input_digest(),
        "results": cs.run_reference(ns),
    }
    path = os.path.join(HERE, "consumer_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    n = json.dumps(out["results"]).count('"hits"')
    print(f"{path}: {n} result lists, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
