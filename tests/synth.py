"""Synthetic inputs shared by tests, the golden generator and bench.py.

Recipe = the reference's benchmark script (tools/benchmark_vectorbase.py:80-94 in
/root/reference): one `default_rng(seed)`; corpus = standard_normal((N, D)) as
float32, rows divided by their L2 norm; the query is drawn *after* the corpus
from the same generator and normalised.  (tests/golden/make_golden.py asserts
that this restatement is byte-identical to the reference script's output.)
"""

from __future__ import annotations

import numpy as np


def make_corpus(n: int, d: int, seed: int) -> tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(seed)
    vectors = rng.standard_normal((n, d)).astype(np.float32)
    norms = np.linalg.norm(vectors, axis=1, keepdims=True)
    vectors /= norms
    query = rng.standard_normal(d).astype(np.float32)
    query /= np.linalg.norm(query)
    return vectors, query


def make_queries(count: int, d: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((count, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def subset_choice(n: int, size: int, seed: int) -> list[int]:
    """tools/benchmark_vectorbase.py:135-136: default_rng(seed).choice(n, size, replace=False).tolist()"""
    return np.random.default_rng(seed).choice(n, size=size, replace=False).tolist()


def explicit_case_arrays(case: dict) -> tuple[np.ndarray, np.ndarray]:
    """(vectors [N, D], query [D]) of an `explicit` golden case (N may be 0)."""
    q = np.asarray(case["query"], dtype=np.float32)
    v = np.asarray(case["vectors"], dtype=np.float32).reshape(-1, q.shape[0])
    return v, q
