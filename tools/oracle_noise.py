#!/usr/bin/env python3
"""How far the reference arithmetic (float32 numpy/OpenBLAS dot, vectorbase.py:176 + :44-47) sits from the exact score, as a function of
the size of the dot product.  Round 3 turned this measurement into a hand-set near-tie width; since round 4 `check_topk_parity` measures the
same thing itself, per query, against a float64 referee (oracle.vectorbase_oracle.scores_f64) -- this script is what to run to see the numbers.  CPU only.

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
        s64 = vo.scores_f64(v, q)
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
    print(f"2^-24 = {2.0 ** -24:.3e}; TIE_EPS (the rule without a referee) = {vo.TIE_EPS:.3e}")


if __name__ == "__main__":
    main()
