#!/usr/bin/env python3
"""Kernel-only timing sweep of the streaming scan's launch geometry (HIP events via
libtavb's profile API).  Writes gpurun_out/sweep_<tag>.json.  Tuning aid, not a test."""

from __future__ import annotations

import argparse
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--tag", default="scan")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()

    import torch

    from bench import host_queries, make_device_corpus
    from typeagent_py_amd import _native

    eng = _native.Engine(0)
    corpus = make_device_corpus(eng, args.rows, args.dim, 1043, args.dtype)
    eng.set_corpus_tensor(corpus)
    qs = host_queries(16, args.dim, 7)
    esize = 2 if args.dtype == "fp16" else 4
    nbytes = args.rows * args.dim * esize
    thr = np.float32(0.0)

    combos = []
    geoms = [(256, 16), (512, 8), (1024, 4), (512, 16), (1024, 8), (2048, 4), (256, 8), (768, 8), (2048, 8)]
    if args.quick:
        geoms = [(256, 16), (512, 8), (1024, 4)]
    for (blocks, waves), unroll, nt, pipe in itertools.product(geoms, (1, 2, 4), (1, 0), (0, 1)):
        if pipe and unroll == 4:
            continue
        combos.append(dict(scan_blocks=blocks, scan_waves=waves, scan_unroll=unroll, scan_nt=nt, scan_pipe=pipe))
    results = []
    eng.profile_enable(True)
    for c in combos:
        for name, val in c.items():
            eng.set_option(name, val)
        try:
            for i in range(2):
                eng.search(qs[i], args.k, thr)
            eng.profile_reset()
            for i in range(args.iters):
                eng.search(qs[i % len(qs)], args.k, thr)
            ms, n = eng.profile_read(_native.KERNEL_SCAN)
            mms, mn = eng.profile_read(_native.KERNEL_MERGE)
            avg = ms / max(n, 1)
            r = dict(c, scan_ms=avg, gbs=nbytes / (avg * 1e-3) / 1e9, merge_us=mms / max(mn, 1) * 1e3)
        except Exception as exc:  # keep sweeping
            r = dict(c, error=str(exc))
        results.append(r)
        print(json.dumps(r), flush=True)
    good = [r for r in results if "gbs" in r]
    good.sort(key=lambda r: -r["gbs"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"sweep_{args.tag}.json"), "w") as f:
        json.dump({"rows": args.rows, "dim": args.dim, "dtype": args.dtype, "k": args.k, "results": results, "best": good[:5]}, f, indent=1)
    print("BEST", json.dumps(good[:5]))


if __name__ == "__main__":
    main()
