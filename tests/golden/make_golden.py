#!/usr/bin/env python3
"""Generate tests/golden/vectorbase_golden.json from the VERBATIM reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every expected result in the JSON is produced by the reference's own
`VectorBase` class (src/typeagent/aitools/vectorbase.py executed unmodified via
oracle/ref_loader.py) -- not by our oracle and not by our engine.  Large inputs
are not stored; they are regenerated from the seed with the recipe of the
reference's benchmark script (tools/benchmark_vectorbase.py:80-94) and pinned
by a sha256 of the corpus bytes so that a drifting RNG is detected, not
silently compared.
"""

from __future__ import annotations

import hashlib
import json
import os
import runpy
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from tests.synth import make_corpus, subset_choice  # noqa: E402


def _hits(res):
    return {"items": [int(r.item) for r in res], "scores": [float(r.score) for r in res]}


def _sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def seeded_cases():
    """(name, N, D, seed, [(kind, kwargs)...])"""
    return [
        ("bench_1k_d1536_seed42", 1_000, 1536, 42, [("full", dict(max_hits=10, min_score=0.0))]),
        (
            "cfg1_10k_d1536_seed43",
            10_000,
            1536,
            43,
            [
                ("full", dict(max_hits=10, min_score=0.0)),
                ("full", dict(max_hits=32, min_score=0.0)),
                ("full", dict(max_hits=50, min_score=0.52)),
                ("full", dict(max_hits=None, min_score=None)),
                ("full", dict(max_hits=25, min_score=0.7)),
                ("full", dict(max_hits=100, min_score=0.535)),
                ("full", dict(max_hits=300, min_score=0.0)),
                ("subset", dict(subset_seed=99, subset_size=1000, max_hits=10, min_score=0.0)),
                ("subset", dict(subset_seed=5, subset_size=64, max_hits=32, min_score=0.5)),
                ("subset", dict(subset_seed=6, subset_size=20, max_hits=32, min_score=0.0)),
            ],
        ),
        ("d384_2k_seed7", 2_000, 384, 7, [("full", dict(max_hits=10, min_score=0.0)), ("full", dict(max_hits=64, min_score=0.5))]),
        ("d100_777_seed8", 777, 100, 8, [("full", dict(max_hits=32, min_score=0.0))]),
        ("d3_500_seed9", 500, 3, 9, [("full", dict(max_hits=32, min_score=0.85)), ("full", dict(max_hits=5, min_score=0.0))]),
        ("d1_50_seed10", 50, 1, 10, [("full", dict(max_hits=10, min_score=0.0))]),
        ("d1536_100k_seed143", 100_000, 1536, 143, [("full", dict(max_hits=32, min_score=0.0)), ("full", dict(max_hits=50, min_score=0.54))]),
        ("cfg2_1m_d1536_seed1043", 1_000_000, 1536, 1043, [("full", dict(max_hits=32, min_score=0.0)), ("full", dict(max_hits=10, min_score=0.0))]),
    ]


def explicit_cases(mod):
    """Small cases stored with their inputs, including the reference's own known-answer tests."""
    f32 = np.float32
    out = []

    def run(name, vectors, query, cite=None, **kw):
        vb = ref_loader.make_reference_vectorbase()
        vecs = np.asarray(vectors, dtype=f32)
        if vecs.size:
            vb.add_embeddings(None, vecs)
        q = np.asarray(query, dtype=f32)
        subset = kw.pop("subset", None)
        pred_mod = kw.pop("predicate_mod", None)
        entry = {"name": name, "vectors": vecs.tolist(), "query": q.tolist(), "args": dict(kw)}
        if cite:
            entry["cite"] = cite
        with np.errstate(invalid="ignore"):
            if subset is not None:
                entry["subset"] = list(subset)
                try:
                    res = vb.fuzzy_lookup_embedding_in_subset(q, list(subset), **kw)
                    entry["expect"] = _hits(res)
                except IndexError as e:
                    entry["raises"] = "IndexError"
            elif pred_mod is not None:
                entry["predicate_mod"] = pred_mod
                res = vb.fuzzy_lookup_embedding(q, predicate=lambda i: i % pred_mod[0] == pred_mod[1], **kw)
                entry["expect"] = _hits(res)
            else:
                res = vb.fuzzy_lookup_embedding(q, **kw)
                entry["expect"] = _hits(res)
        out.append(entry)

    # reference tests/test_vectorbase.py:239-252
    run("ref_test_normalized_score_scale", [[1, 0], [0, 1], [-1, 0]], [1, 0],
        cite="tests/test_vectorbase.py:239-252", max_hits=3, min_score=0.0)
    # reference tests/test_benchmark_embeddings.py:229-249 (two rows, query [0,1])
    run("ref_test_benchmark_embeddings_two_rows", [[1, 0], [0, 1]], [0, 1],
        cite="tests/test_benchmark_embeddings.py:229-277", max_hits=2, min_score=0.0)
    # sample embeddings of tests/test_vectorbase.py:62-69 (un-normalised rows -> plain dot products)
    samp = [[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.7, 0.8, 0.9]]
    run("ref_test_samples_full", samp, samp[0], cite="tests/test_vectorbase.py:62-69", max_hits=None, min_score=None)
    run("ref_test_samples_subset_all", samp, samp[0], cite="tests/test_vectorbase.py:209-236", subset=[0, 1, 2])
    run("ref_test_samples_subset_one", samp, samp[0], cite="tests/test_vectorbase.py:209-236", subset=[1])
    run("ref_test_samples_subset_empty", samp, samp[0], cite="tests/test_vectorbase.py:209-236", subset=[])
    # empty corpus
    run("empty_corpus", np.zeros((0, 3)), [1, 0, 0], max_hits=5, min_score=0.0)
    # zero row scores exactly 0.5; NaN row vanishes; clip above 1 / below 0
    run("zero_nan_clip_rows", [[0, 0], [np.nan, 1], [3, 0], [-3, 0], [0.5, 0]], [1, 0], max_hits=10, min_score=0.0)
    # min_score edge: 1.0 keeps exact matches, >1 gives nothing
    run("min_score_one", [[1, 0], [0.999, 0], [0, 1]], [1, 0], max_hits=10, min_score=1.0)
    run("min_score_above_one", [[1, 0], [0, 1]], [1, 0], max_hits=10, min_score=1.5)
    # float32 threshold rule: score 0.85 exactly-as-f32 passes min_score=0.85 (python float)
    c = float(np.float32(0.85)) * 2 - 1
    run("threshold_is_float32", [[c, 0], [c - 1e-6, 0]], [1, 0], max_hits=10, min_score=0.85)
    # M <= k branch, k > N
    run("fewer_survivors_than_k", [[1, 0], [0.6, 0.8], [0, 1], [-1, 0]], [1, 0], max_hits=50, min_score=0.5)
    # max_hits = 0 quirk: everything, sorted
    run("max_hits_zero_quirk", [[0.1, 0], [0.9, 0], [0.5, 0], [0.3, 0]], [1, 0], max_hits=0, min_score=0.0)
    # max_hits = 1
    run("max_hits_one", [[0.1, 0], [0.9, 0], [0.5, 0], [0.3, 0]], [1, 0], max_hits=1, min_score=0.0)
    # subset: duplicates, negative ordinal, unsorted, out-of-range
    rows = [[0.1, 0], [0.9, 0], [0.5, 0], [0.3, 0], [0.7, 0]]
    run("subset_duplicates_unsorted", rows, [1, 0], subset=[4, 1, 1, 0], max_hits=10, min_score=0.0)
    run("subset_negative_ordinal", rows, [1, 0], subset=[-1, 2], max_hits=10, min_score=0.0)
    run("subset_out_of_range", rows, [1, 0], subset=[0, 5], max_hits=10, min_score=0.0)
    run("subset_min_score", rows, [1, 0], subset=[0, 1, 2, 3, 4], max_hits=2, min_score=0.7)
    # predicate path (vectorbase.py:191-201): stable sort, ascending ordinal among equal scores
    run("predicate_even", rows + rows, [1, 0], predicate_mod=[2, 0], max_hits=3, min_score=0.0)
    run("predicate_none_pass", rows, [1, 0], predicate_mod=[7, 6], max_hits=3, min_score=0.0)
    return out


def main() -> None:
    mod = ref_loader.load_reference_vectorbase()
    # the reference's own corpus recipe, executed verbatim, must equal tests/synth.make_corpus
    ref_loader._install_stubs()
    sys.modules["typeagent.aitools.vectorbase"] = mod
    try:
        bench = runpy.run_path(os.path.join(ref_loader.REFERENCE_ROOT, "tools", "benchmark_vectorbase.py"))
    finally:
        del sys.modules["typeagent.aitools.vectorbase"]
    vb_ref, q_ref = bench["make_vectorbase"](1_000, 1536, 42)
    v_mine, q_mine = make_corpus(1_000, 1536, 42)
    assert np.array_equal(vb_ref.serialize(), v_mine) and np.array_equal(q_ref, q_mine), "synth recipe drifted"

    golden = {
        "generator": "tests/golden/make_golden.py",
        "reference": "microsoft/typeagent-py src/typeagent/aitools/vectorbase.py (verbatim via oracle/ref_loader.py)",
        "numpy": np.__version__,
        "settings_defaults": {},
        "seeded": [],
        "explicit": explicit_cases(mod),
    }
    # settings known answers (tests/test_vectorbase.py:280-325)
    class _M:
        def __init__(self, n):
            self.model_name = n
    for name in ["text-embedding-3-large", "text-embedding-3-small", "text-embedding-ada-002", "custom-embedding-model"]:
        s = mod.TextEmbeddingIndexSettings(embedding_model=_M(name))
        golden["settings_defaults"][name] = {"min_score": s.min_score, "max_matches": s.max_matches, "batch_size": s.batch_size}

    for name, n, d, seed, runs in seeded_cases():
        print("generating", name, flush=True)
        vectors, query = make_corpus(n, d, seed)
        vb = ref_loader.make_reference_vectorbase(vectors)
        entry = {"name": name, "n": n, "d": d, "seed": seed, "corpus_sha256": _sha(vectors), "query_sha256": _sha(query), "runs": []}
        for kind, kw in runs:
            kw = dict(kw)
            if kind == "full":
                res = vb.fuzzy_lookup_embedding(query, **kw)
                entry["runs"].append({"kind": "full", "args": kw, "expect": _hits(res)})
            else:
                sub = subset_choice(n, kw.pop("subset_size"), kw.pop("subset_seed"))
                res = vb.fuzzy_lookup_embedding_in_subset(query, sub, **kw)
                entry["runs"].append({"kind": "subset", "subset_sha256": _sha(np.asarray(sub, dtype=np.int64)),
                                      "subset_args": {"size": len(sub)}, "args": kw, "expect": _hits(res), "subset": sub if len(sub) <= 64 else None})
        # fp16-rounded variant of the same corpus (what the f16 device corpus holds): oracle
        # consumes the fp16 values upcast to f32 (last paragraph)
        if n <= 100_000 and d == 1536:
            v16 = vectors.astype(np.float16).astype(np.float32)
            q16 = query.astype(np.float16).astype(np.float32)
            vb16 = ref_loader.make_reference_vectorbase(v16)
            entry["runs"].append({"kind": "full_f16_corpus_f32_query", "args": dict(max_hits=32, min_score=0.0),
                                  "expect": _hits(vb16.fuzzy_lookup_embedding(query, max_hits=32, min_score=0.0))})
            entry["runs"].append({"kind": "full_f16_corpus_f16_query", "args": dict(max_hits=32, min_score=0.0),
                                  "expect": _hits(vb16.fuzzy_lookup_embedding(q16, max_hits=32, min_score=0.0))})
        golden["seeded"].append(entry)
        del vb, vectors

    # fix-up: remember the subset seeds (popped above) for regeneration
    for (name, n, d, seed, runs), entry in zip(seeded_cases(), golden["seeded"]):
        j = 0
        for kind, kw in runs:
            if kind == "subset":
                entry["runs"][j]["subset_args"]["seed"] = kw["subset_seed"]
            j += 1

    path = os.path.join(HERE, "vectorbase_golden.json")
    with open(path, "w") as f:
        json.dump(golden, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
