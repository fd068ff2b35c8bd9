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
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    thr = np.float32(ms)
    print("rows", rows, "k", k, "min_score", ms)
    print("host-synchronous search, ONE launch (lists -> pinned, host merge; the default): %.1f us" % med(lambda: eng.search(q, k, thr)))
    direct = eng.get_option("last_direct") >= 1  # (corpora up to small_direct_bytes = 128 MiB)
    print("   (query inside the kernel arguments: %s)" % ("yes" if eng.get_option("last_direct") == 2 else "no"))
    eng.set_option("inline_query", 0)
    print("   ... with the query copied to a device buffer first (inline_query=0):          %.1f us" % med(lambda: eng.search(q, k, thr)))
    eng.set_option("inline_query", 1)
    for waves in (2, 4, 8, 16):
        eng.set_option("scan_waves", waves)
        print("   ... with scan_waves=%-2d:                                                       %.1f us" % (waves, med(lambda: eng.search(q, k, thr))))
    eng.set_option("scan_waves", 16)
    for blocks in ((8, 16, 32, 64, 128, 256) if direct else ()):
        eng.set_option("scan_blocks", blocks)
        print("   ... with scan_blocks=%-3d (lists merged on the host; capped by small_direct_keys / k):  %.1f us" % (blocks, med(lambda: eng.search(q, k, thr))))
    eng.set_option("scan_blocks", 0)
    eng.set_option("small_direct_bytes", 0)
    print("host-synchronous search, two launches (scan + device merge -> pinned; round 3): %.1f us" % med(lambda: eng.search(q, k, thr)))
    eng.set_option("graph_max_bytes", 256 << 20)
    print("the same as ONE captured HIP graph replay (graph_max_bytes option):             %.1f us" % med(lambda: eng.search(q, k, thr)))
    assert eng.get_option("last_graph") == 1
    eng.set_option("graph_max_bytes", 0)
    print("device-resident search_device + synchronize:                                    %.1f us" % med(lambda: (eng.search_device(dq, k, ms), eng.synchronize())))
    print("synchronize only:                                                                %.1f us" % med(lambda: eng.synchronize()))
    print("get_option (ctypes round trip):                                                  %.1f us" % med(lambda: eng.get_option("last_tier")))
    eng.set_option("small_direct_bytes", 128 << 20)
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(200):
        eng.search(q, k, thr)
    for kid, name in ((_native.KERNEL_SCAN, "scan"), (_native.KERNEL_MERGE, "merge")):
        ms, n = eng.profile_read(kid)
        print("  kernel %-6s %.1f us avg over %d launches" % (name, ms / max(n, 1) * 1e3, n))
    print("with profiling on (adds event records): %.1f us" % med(lambda: eng.search(q, k, thr)))
    # the class-level call (what a typeagent consumer pays): VectorBase.fuzzy_lookup_embedding -> list[ScoredInt]
    from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase

    class _Null:
        model_name = "bench"

    vb = VectorBase(TextEmbeddingIndexSettings(_Null()), device=0)
    vb.adopt_device_corpus(corpus)
    print("VectorBase.fuzzy_lookup_embedding -> list[ScoredInt]:                            %.1f us" % med(lambda: vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=ms)))


if __name__ == "__main__":
    main()
