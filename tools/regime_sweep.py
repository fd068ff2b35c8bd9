#!/usr/bin/env python3
"""Looks for regime cliffs: time of one batched lookup (device-resident queries, tavb_search_device + tavb_synchronize) over a grid of
corpus sizes x batch sizes, per dtype, as a markdown table -- and every cell that is SLOWER than a cell with more rows and at least as
many queries (or more queries and at least as many rows) by more than 10 %: more work should not take less time.

    python tools/regime_sweep.py [--dtype fp16,fp32] [--rows 1000,10000,...] [--sizes 1,2,...] [--dim 1536] [--k 32]
"""
import argparse
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16,fp32")
    ap.add_argument("--rows", default="1000,5000,20000,30720,50000,82000,120000,165000,250000,500000,1000000,3000000")
    ap.add_argument("--sizes", default="1,2,3,4,5,8,16,32,33,64,65,128,129,256,257,512,1024,2048")
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import torch

    rows_list = [int(x) for x in args.rows.split(",")]
    sizes = [int(x) for x in args.sizes.split(",")]
    queries = torch.from_numpy(bench.host_queries(max(sizes), args.dim, 77)).cuda()
    for dtype in args.dtype.split(","):
        eng = _native.Engine(0)
        for o in args.opt:
            n, v = o.split("=")
            eng.set_option(n, int(v))
        big = bench.make_device_corpus(eng, max(rows_list), args.dim, 1043, dtype)
        ms, tier = {}, {}
        for rows in rows_list:
            eng.set_corpus_tensor(big[:rows])
            for nq in sizes:
                q = queries[:nq]
                for _ in range(3):
                    eng.search_device(q, args.k, 0.0)
                eng.synchronize()
                best = float("inf")
                for _ in range(3):  # best of three short loops: a stall of the box is not a cliff of the library
                    reps = max(3, min(30, int(2e-3 / max(best if best < 1 else 1e-4, 2e-5))))
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        eng.search_device(q, args.k, 0.0)
                    eng.synchronize()
                    best = min(best, (time.perf_counter() - t0) / reps)
                ms[(rows, nq)] = best * 1e3
                tier[(rows, nq)] = eng.get_option("last_tier")
        print(f"\n## {dtype}, D = {args.dim}, k = {args.k}: ms per lookup (tier: 1-3 streaming scan, 4 wide tile, 5 32/64-query tile)\n")
        print("| rows \\ queries | " + " | ".join(str(n) for n in sizes) + " |")
        print("|---|" + "---|" * len(sizes))
        for rows in rows_list:
            print(f"| {rows} | " + " | ".join(f"{ms[(rows, n)]:.3f} ({tier[(rows, n)]})" for n in sizes) + " |")
        cliffs = []
        for (r, n), t in ms.items():
            for (r2, n2), t2 in ms.items():
                if (r2, n2) != (r, n) and r2 >= r and n2 >= n and t > 1.10 * t2 and t - t2 > 0.01:
                    cliffs.append((t / t2, r, n, t, r2, n2, t2))
        cliffs.sort(reverse=True)
        print(f"\ncells slower than a cell with at least as many rows AND queries by more than 10 % (and 10 us): {len(cliffs)}")
        for ratio, r, n, t, r2, n2, t2 in cliffs[:40]:
            print(f"  {r} rows x {n} queries: {t:.3f} ms (tier {tier[(r, n)]})  >  {r2} rows x {n2} queries: {t2:.3f} ms (tier {tier[(r2, n2)]})  [{ratio:.2f}x]")
        del big
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
