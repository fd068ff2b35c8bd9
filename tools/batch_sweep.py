#!/usr/bin/env python3
"""Throughput of one batched lookup as a function of the batch size (streaming tiers vs the MFMA kernel).
usage: python tools/batch_sweep.py [--rows N] [--dtype fp32|fp16] [--opt name=value ...]"""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--sizes", default="1,2,4,8,16,24,32,64,128,256")
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import torch

    eng = _native.Engine(0)
    for o in args.opt:
        n, v = o.split("=")
        eng.set_option(n, int(v))
    corpus = bench.make_device_corpus(eng, args.rows, 1536, 1043, args.dtype)
    eng.set_corpus_tensor(corpus)
    bytes_per_pass = args.rows * 1536 * (2 if args.dtype == "fp16" else 4)
    out = []
    for nq in [int(x) for x in args.sizes.split(",")]:
        q = torch.from_numpy(bench.host_queries(nq, 1536, 7 + nq)).cuda()
        for _ in range(3):
            eng.search_device(q, args.k, 0.0)
        torch.cuda.synchronize()
        eng.synchronize()
        reps = 20 if nq <= 32 else 8
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.search_device(q, args.k, 0.0)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / reps
        row = dict(nq=nq, ms=round(dt * 1e3, 3), qps=round(nq / dt, 1), corpus_passes_TBps=round(bytes_per_pass / dt / 1e12, 3), tier=eng.get_option("last_tier"))
        print(json.dumps(row), flush=True)
        out.append(row)


if __name__ == "__main__":
    main()
