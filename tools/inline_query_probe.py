"""Why does a one-launch lookup with the query inside the kernel arguments read 29 us in tools/latency_breakdown.py and 79 us in bench.py's cfg1 loop?
Blocks of 2000 host-synchronous lookups on 10k x 1536 fp32: the same query every call, a rotation of 64 queries, with inline_query on and off."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def block(f, n=2000):
    t = np.empty(n)
    for i in range(n):
        a = time.perf_counter_ns()
        f(i)
        t[i] = (time.perf_counter_ns() - a) / 1e3
    return "p50 %.1f  p90 %.1f  p99 %.1f  max %.0f" % (np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), t.max())


def main():
    eng = _native.Engine(0)
    corpus = bench.make_device_corpus(eng, 10_000, 1536, 50_041, "fp32")
    eng.set_corpus_tensor(corpus)
    qs = bench.host_queries(64, 1536, 4242)
    q0 = qs[0].copy()
    thr = np.float32(0.0)
    for inline in (1, 0, 1):
        eng.set_option("inline_query", inline)
        print("inline_query =", inline)
        for rep in range(3):
            print("   same query      :", block(lambda i: eng.search(q0, 10, thr)))
        for rep in range(3):
            print("   rotating queries:", block(lambda i: eng.search(qs[i % 64], 10, thr)))
        print("   same query again:", block(lambda i: eng.search(q0, 10, thr)))


if __name__ == "__main__":
    main()
