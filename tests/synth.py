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


def make_clustered_corpus(n: int, d: int, seed: int, cluster_rows: int = 100, spread: float = 0.002, n_queries: int = 0):
    """Clustered twin of make_corpus (the host-side restatement of bench.py's `kind="clustered"` recipe, scaled down): n // cluster_rows
    unit centres; row i belongs to cluster (i * 7368787) % n_clusters and is centre + spread * noise / sqrt(d), normalised; every row with
    i % 8 == 5 takes the NEXT cluster's centre as its noise -- those rows of a cluster are exact duplicates of one another.  Queries sit
    next to centres (centre + 0.05 * noise / sqrt(d)): their top hits are one cluster, with scores packed inside ~spread / 10.
    Returns (vectors [n, d] float32, queries [n_queries, d] float32, cluster id of every row, cluster id of every query)."""
    rng = np.random.default_rng(seed)
    n_c = max(1, n // cluster_rows)
    centres = rng.standard_normal((n_c, d)).astype(np.float32)
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    ids = np.arange(n, dtype=np.int64)
    cl = (ids * 7_368_787) % n_c
    v = rng.standard_normal((n, d)).astype(np.float32) * np.float32(spread / np.sqrt(d))
    dup = (ids % 8) == 5
    v[dup] = centres[(cl[dup] + 1) % n_c] * np.float32(spread)
    v += centres[cl]
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    qc = rng.choice(n_c, size=n_queries, replace=n_c < n_queries) if n_queries else np.zeros(0, dtype=np.int64)
    q = centres[qc] + np.float32(0.05 / np.sqrt(d)) * rng.standard_normal((n_queries, d)).astype(np.float32)
    if n_queries:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    return v.astype(np.float32), q.astype(np.float32), cl, qc
