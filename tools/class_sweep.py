#!/usr/bin/env python3
"""The drop-in class next to the engine: `VectorBase.fuzzy_lookup_embedding(s)` (host queries in, `list[ScoredInt]` out) against
`Engine.search_device` (device-resident queries, keys out) over corpus sizes x batch sizes -- where the Python layer costs more than it
should -- plus the forms a caller reaches through the class only: per-query thresholds, `min_score` 0.85, the predicate branch, appends
between lookups.

    python tools/class_sweep.py [--dtype fp32] [--dim 1536]
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase  # noqa: E402


class _Null:
    model_name = "sweep"


def best_of(fn, loops=3):
    best = float("inf")
    for _ in range(loops):
        reps = max(3, min(30, int(3e-3 / max(best if best < 1 else 1e-4, 2e-5))))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--rows", default="1000,10000,100000,1000000")
    ap.add_argument("--sizes", default="1,4,16,64,256,1024")
    args = ap.parse_args()
    import torch

    rows_list = [int(x) for x in args.rows.split(",")]
    sizes = [int(x) for x in args.sizes.split(",")]
    hq = bench.host_queries(max(sizes), args.dim, 33)
    dq = torch.from_numpy(hq).cuda()
    for k in (10, 50):
        print(f"\n## {args.dtype}, D = {args.dim}, k = {k}: class ms / engine ms per call (class = host queries in, list[list[ScoredInt]] out; engine = device queries, keys out)\n")
        print("| rows \\ queries | " + " | ".join(str(n) for n in sizes) + " |\n|---|" + "---|" * len(sizes))
        for rows in rows_list:
            vb = VectorBase(TextEmbeddingIndexSettings(_Null()), device=0)
            corpus = bench.make_device_corpus(vb.engine, rows, args.dim, 1043, args.dtype)
            vb.adopt_device_corpus(corpus)
            eng = vb.engine
            cells = []
            for nq in sizes:
                if nq == 1:
                    c_ms = best_of(lambda: vb.fuzzy_lookup_embedding(hq[0], max_hits=k, min_score=0.0))
                else:
                    c_ms = best_of(lambda: vb.fuzzy_lookup_embeddings(hq[:nq], max_hits=k, min_score=0.0))
                q = dq[:nq]

                def e():
                    eng.search_device(q, k, 0.0)
                    eng.synchronize()
                e_ms = best_of(e)
                cells.append(f"{c_ms:.3f} / {e_ms:.3f}" + (" **" if c_ms > 1.5 * e_ms + 0.03 else ""))
            print(f"| {rows} | " + " | ".join(cells) + " |", flush=True)
            if rows == rows_list[-1] and k == 10:
                nq = 64
                thr = np.linspace(0.0, 0.6, nq).tolist()
                print(f"\nforms on {rows} rows, {nq} queries, k = {k}:")
                print(f"  uniform min_score 0.0        {best_of(lambda: vb.fuzzy_lookup_embeddings(hq[:nq], max_hits=k, min_score=0.0)):.3f} ms")
                print(f"  uniform min_score 0.85       {best_of(lambda: vb.fuzzy_lookup_embeddings(hq[:nq], max_hits=k, min_score=0.85)):.3f} ms")
                print(f"  one min_score per query      {best_of(lambda: vb.fuzzy_lookup_embeddings(hq[:nq], max_hits=k, min_score=thr)):.3f} ms")
                print(f"  {nq} sequential single lookups {best_of(lambda: [vb.fuzzy_lookup_embedding(hq[i], max_hits=k, min_score=0.0) for i in range(nq)]):.3f} ms")
                print(f"  predicate (even ordinals), one query, min_score 0.55   {best_of(lambda: vb.fuzzy_lookup_embedding(hq[0], max_hits=k, min_score=0.55, predicate=lambda i: i % 2 == 0)):.3f} ms")
                print(f"  predicate (even ordinals), one query, min_score 0.0    {best_of(lambda: vb.fuzzy_lookup_embedding(hq[0], max_hits=k, min_score=0.0, predicate=lambda i: i % 2 == 0), loops=1):.3f} ms")
            del vb, corpus
            torch.cuda.empty_cache()
    # appends between lookups (host-owned corpus: the reference re-copies the whole matrix per append)
    vb = VectorBase(TextEmbeddingIndexSettings(_Null()), device=0)
    rng = np.random.default_rng(3)
    base = rng.standard_normal((50_000, args.dim)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    vb.add_embeddings(None, base)
    vb.fuzzy_lookup_embedding(hq[0], max_hits=10, min_score=0.0)
    t0 = time.perf_counter()
    for i in range(200):
        vb.add_embedding(None, hq[i % len(hq)])
        vb.fuzzy_lookup_embedding(hq[0], max_hits=10, min_score=0.0)
    print(f"\nappend one row + one lookup, 200 times on a 50k-row host-owned corpus: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per pair")
    t0 = time.perf_counter()
    for i in range(50):
        vb.add_embedding(None, hq[i % len(hq)])
        vb.fuzzy_lookup_embeddings(hq[:128], max_hits=10, min_score=0.0)
    print(f"append one row + one 128-query batch, 50 times: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per pair")


if __name__ == "__main__":
    main()
