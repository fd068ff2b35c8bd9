#!/usr/bin/env python3
"""Where the microseconds of a small-corpus lookup go (cfg1 scale: 10k x 1536 fp32, top-10)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def med(f, n=400, warm=50):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter_ns()
        f()
        ts.append(time.perf_counter_ns() - t0)
    return float(np.median(ts)) / 1e3


def main():
    import torch

    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    eng = _native.Engine(0)
    corpus = bench.make_device_corpus(eng, rows, 1536, 43, "fp32")
    eng.set_corpus_tensor(corpus)
    q = bench.host_queries(1, 1536, 7)[0]
    dq = torch.from_numpy(q[None, :]).cuda()
    thr = np.float32(0.0)
    print("rows", rows)
    print("host-synchronous search (H2D + scan + merge->pinned + sync + decode): %.1f us" % med(lambda: eng.search(q, 10, thr)))
    eng.set_option("graph_max_bytes", 256 << 20)
    print("the same as ONE captured HIP graph replay (graph_max_bytes option):   %.1f us" % med(lambda: eng.search(q, 10, thr)))
    assert eng.get_option("last_graph") == 1
    eng.set_option("graph_max_bytes", 0)
    print("device-resident search_device + synchronize:                          %.1f us" % med(lambda: (eng.search_device(dq, 10, 0.0), eng.synchronize())))
    print("synchronize only:                                                      %.1f us" % med(lambda: eng.synchronize()))
    print("get_option (ctypes round trip):                                        %.1f us" % med(lambda: eng.get_option("last_tier")))
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(200):
        eng.search(q, 10, thr)
    for kid, name in ((_native.KERNEL_SCAN, "scan"), (_native.KERNEL_MERGE, "merge")):
        ms, n = eng.profile_read(kid)
        print("  kernel %-6s %.1f us avg over %d launches" % (name, ms / max(n, 1) * 1e3, n))
    print("with profiling on (adds event records): %.1f us" % med(lambda: eng.search(q, 10, thr)))


if __name__ == "__main__":
    main()
