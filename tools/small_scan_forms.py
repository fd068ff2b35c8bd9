"""Scan-kernel forms on a small corpus (10k / 1294 x 1536 fp32, one query, one launch): rows in flight per wave, software prefetch, non-temporal loads,
waves per workgroup -- with the query copied first (inline_query=0: every form has that variant), and the default form with the query in the kernel arguments."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def med(f, n=600):
    for _ in range(100):
        f()
    t = np.empty(n)
    for i in range(n):
        a = time.perf_counter_ns()
        f()
        t[i] = (time.perf_counter_ns() - a) / 1e3
    return float(np.median(t))


def main():
    eng = _native.Engine(0)
    q = bench.host_queries(1, 1536, 7)[0]
    thr = np.float32(0.0)
    for rows, k in ((10_000, 10), (1294, 50)):
        corpus = bench.make_device_corpus(eng, rows, 1536, 50_041, "fp32")
        eng.set_corpus_tensor(corpus)
        print("rows", rows, "k", k)
        eng.set_option("inline_query", 1)
        print("   default form, query in the kernel arguments: %.1f us" % med(lambda: eng.search(q, k, thr)))
        eng.set_option("inline_query", 0)
        for unroll, nt, pipe in ((2, 1, 0), (1, 1, 0), (4, 1, 0), (2, 0, 0), (1, 0, 0), (4, 0, 0), (2, 1, 1), (1, 1, 1), (2, 0, 1), (1, 0, 1)):
            eng.set_option("scan_unroll", unroll)
            eng.set_option("scan_nt", nt)
            eng.set_option("scan_pipe", pipe)
            print("   query copied first, unroll=%d nt=%d pipe=%d:        %.1f us" % (unroll, nt, pipe, med(lambda: eng.search(q, k, thr))))
        eng.set_option("scan_unroll", 2)
        eng.set_option("scan_nt", 1)
        eng.set_option("scan_pipe", 0)


if __name__ == "__main__":
    main()
