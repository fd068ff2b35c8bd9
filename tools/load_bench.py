#!/usr/bin/env python3
"""Load-path throughput: float32 host rows -> device corpus through tavb_upload_rows (pinned double-buffered
staging, async H2D, on-device fp16 conversion), against the PCIe Gen5 x16 spec rate (63 GB/s) and against what the
round-1 path did (numpy astype + pageable torch copy).  Prints one JSON line per case."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from typeagent_py_amd import _native

    eng = _native.Engine(0)
    rows, dim = 1_000_000, 1536
    host = np.random.default_rng(0).standard_normal((rows, dim)).astype(np.float32)
    gb = host.nbytes / 1e9
    for dtype, name in ((_native.TAVB_F32, "fp32"), (_native.TAVB_F16, "fp16")):
        eng.corpus = None
        eng.upload_rows(host[:1000], 0, dtype, capacity_hint=rows)  # allocation + first touch outside the timing
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            eng.upload_rows(host, 0, dtype)
            times.append(time.perf_counter() - t0)
        t = min(times)
        print(json.dumps({"path": "tavb_upload_rows", "corpus_dtype": name, "rows": rows, "dim": dim, "host_GB": gb, "seconds": t,
                          "host_GBps": gb / t, "pcie_spec_GBps": 63.0, "frac_of_pcie": gb / t / 63.0}))
        # round-1 form for comparison
        dst = torch.empty((rows, dim), dtype=torch.float16 if name == "fp16" else torch.float32, device="cuda")
        t0 = time.perf_counter()
        src = host.astype(np.float16) if name == "fp16" else host
        dst.copy_(torch.from_numpy(src))
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
        print(json.dumps({"path": "round-1: numpy astype + pageable torch copy", "corpus_dtype": name, "seconds": t1, "host_GBps": gb / t1}))


if __name__ == "__main__":
    main()
