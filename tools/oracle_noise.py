#!/usr/bin/env python3
"""How far the reference arithmetic (float32 numpy/OpenBLAS dot, vectorbase.py:176 + :44-47) sits from the exact score, as a function of
the size of the dot product: the measurement behind oracle.vectorbase_oracle.tie_eps_at().  CPU only.

    python tools/oracle_noise.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vectorbase_oracle as vo  # noqa: E402
from tests.synth import make_clustered_corpus, make_corpus, make_queries  # noqa: E402


def noise(v, qs, top=64):
    errs = []
    for q in qs:
        s32 = vo.scores_full(v, q)
        s64 = np.clip((v.astype(np.float64) @ q.astype(np.float64) + 1.0) / 2.0, 0.0, 1.0)
        best = np.argsort(-s64)[:top]
        errs.append(np.abs(s32[best] - s64[best]))
    return np.concatenate(errs), float(np.mean([np.sort(vo.scores_full(v, q))[-1] for q in qs[:8]]))


def main():
    v, q, _, _ = make_clustered_corpus(60_000, 1536, 4200, cluster_rows=100, n_queries=64)
    e, top = noise(v.astype(np.float16).astype(np.float32), q)
    print(f"clustered corpus (best score ~{top:.4f}): |f32 - f64| max {e.max():.3e}  p99 {np.percentile(e, 99):.3e}  median {np.median(e):.3e}")
    vg, _ = make_corpus(60_000, 1536, 1)
    e, top = noise(vg, make_queries(64, 1536, 2))
    print(f"isotropic corpus (best score ~{top:.4f}): |f32 - f64| max {e.max():.3e}  p99 {np.percentile(e, 99):.3e}  median {np.median(e):.3e}")
    print(f"2^-24 = {2.0 ** -24:.3e}; tie_eps_at(0.57) = {vo.tie_eps_at(0.57):.3e}, tie_eps_at(0.999) = {vo.tie_eps_at(0.999):.3e}")


if __name__ == "__main__":
    main()
