#!/usr/bin/env python3
"""Two more sweeps in the spirit of tools/regime_sweep.py (more work must not take less time):
  * the subset form (`tavb_search_subset_resident`, one query) over subset sizes on one corpus, next to the full scan;
  * `k` (1 .. 256) for one query and for batches of 32 / 128 / 1024 queries.

    python tools/misc_sweep.py [--rows 3000000] [--dtype fp16]
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def best_of(fn, sync, loops=3):
    best = float("inf")
    for _ in range(loops):
        reps = max(3, min(30, int(2e-3 / max(best if best < 1 else 1e-4, 2e-5))))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=3_000_000)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--dim", type=int, default=1536)
    args = ap.parse_args()
    import torch

    eng = _native.Engine(0)
    corpus = bench.make_device_corpus(eng, args.rows, args.dim, 1043, args.dtype)
    eng.set_corpus_tensor(corpus)
    hq = bench.host_queries(1024, args.dim, 91)
    dq = torch.from_numpy(hq).cuda()
    thr = np.float32(0.0)
    rng = np.random.default_rng(5)
    print(f"## subset form, one query, {args.rows} x {args.dim} {args.dtype} rows, k = 32: ms per lookup (host call incl. the synchronise)\n")
    print("| subset size | ms | tier |\n|---|---|---|")
    prev = None
    for s in (100, 1000, 10_000, 50_000, 100_000, 300_000, 1_000_000, args.rows):
        rows = np.sort(rng.choice(args.rows, size=s, replace=False)).astype(np.int32) if s < args.rows else np.arange(args.rows, dtype=np.int32)
        dev_rows = eng.rows_to_device(rows)
        for _ in range(3):
            eng.search_subset_resident(hq[0], dev_rows, 32, thr)
        ms = best_of(lambda: eng.search_subset_resident(hq[0], dev_rows, 32, thr), lambda: None)
        flag = "  <-- slower than a bigger subset?" if False else ""
        print(f"| {s} | {ms:.3f} | {eng.get_option('last_tier')} |{flag}")
    for _ in range(3):
        eng.search(hq[0], 32, thr)
    print(f"| full scan (`tavb_search`) | {best_of(lambda: eng.search(hq[0], 32, thr), lambda: None):.3f} | {eng.get_option('last_tier')} |")
    print(f"\n## k sweep on the same corpus: ms per lookup (device-resident queries)\n")
    ks = (1, 10, 32, 50, 64, 65, 100, 128, 256)
    print("| queries \\ k | " + " | ".join(str(k) for k in ks) + " |\n|---|" + "---|" * len(ks))
    for nq in (1, 4, 32, 128, 1024):
        cells = []
        for k in ks:
            q = dq[:nq]
            for _ in range(3):
                eng.search_device(q, k, 0.0)
            eng.synchronize()
            cells.append(f"{best_of(lambda: eng.search_device(q, k, 0.0), eng.synchronize):.3f} ({eng.get_option('last_tier')})")
        print(f"| {nq} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
