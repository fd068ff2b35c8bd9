"""Host-synchronous lookups of a FEW queries at once on a small fp32 corpus (batched related-term lookups: adapters.install_batched_lookup_terms):
median us per call of Engine.search_batch, and per query, against nq sequential single lookups."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def med(f, n=400):
    for _ in range(60):
        f()
    t = np.empty(n)
    for i in range(n):
        a = time.perf_counter_ns()
        f()
        t[i] = (time.perf_counter_ns() - a) / 1e3
    return float(np.median(t))


def main():
    eng = _native.Engine(0)
    qs = bench.host_queries(64, 1536, 7)
    for dtype in ("fp32", "fp16"):
        for rows in (1294, 10_000):
            corpus = bench.make_device_corpus(eng, rows, 1536, 50_041, dtype)
            eng.set_corpus_tensor(corpus)
            for k, ms in ((50, 0.85), (10, 0.0)):
                thr = np.float32(ms)
                one = med(lambda: eng.search(qs[0], k, thr))
                line = ["%s %5d rows k=%-2d @%.2f: 1 query %.1f us |" % (dtype, rows, k, ms, one)]
                for nq in (2, 4, 8, 16, 32):
                    t = med(lambda: eng.search_batch(qs[:nq], k, thr))
                    line.append("%d: %.1f (tier %d, direct %d)" % (nq, t, eng.get_option("last_tier"), eng.get_option("last_direct")))
                print(" ".join(line), flush=True)


if __name__ == "__main__":
    main()
