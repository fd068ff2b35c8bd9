#!/usr/bin/env python3
"""How long does the HOST take to submit one batched lookup (tavb_search_device returns after enqueueing), next to the time the GPU needs for it?
A batch whose submission takes longer than its early kernels leaves the GPU idle between them.

    python tools/host_submit_time.py [rows] [dim] [fp32|fp16] [queries]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def main():
    import torch

    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
    dtype = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    nq = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    eng = _native.Engine(0)
    corpus = bench.make_device_corpus(eng, rows, dim, 43, dtype)
    eng.set_corpus_tensor(corpus)
    dq = torch.from_numpy(bench.host_queries(nq, dim, 7)).cuda()
    keys = torch.empty((nq, 32), dtype=torch.int64, pin_memory=True)
    for _ in range(20):
        eng.search_device(dq, 32, 0.0, out_keys=keys)
    eng.synchronize()
    sub, tot = [], []
    for _ in range(200):
        t0 = time.perf_counter_ns()
        eng.search_device(dq, 32, 0.0, out_keys=keys)
        t1 = time.perf_counter_ns()
        eng.synchronize()
        t2 = time.perf_counter_ns()
        sub.append((t1 - t0) / 1e3)
        tot.append((t2 - t0) / 1e3)
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(50):
        eng.search_device(dq, 32, 0.0, out_keys=keys)
    eng.synchronize()
    kern = sum(eng.profile_read(i)[0] for i in range(9)) / 50 * 1e3
    print(f"{rows} x {dim} {dtype}, {nq} queries: submit {np.median(sub):.1f} us (p90 {np.percentile(sub, 90):.1f}), submit + synchronize {np.median(tot):.1f} us, "
          f"sum of kernel times (events) {kern:.1f} us, tier {eng.get_option('last_tier')}")


if __name__ == "__main__":
    main()
