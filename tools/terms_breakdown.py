#!/usr/bin/env python3
"""A batched related-terms lookup at the reference's scale (1294 x 1536 fp32 rows, 8 / 32 / 64 terms) through `tavb_search_batch`: us per call
and the scan kernel's share, on a gaussian corpus (nothing survives min_score 0.85: empty lists) and on the real-embedding-like one (most rows
survive: full lists, the host merge has work to do), k = 10 / 50.

    python tools/terms_breakdown.py
"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from typeagent_py_amd import _native
eng = _native.Engine(0)
def med(f, n=300):
    for _ in range(30): f()
    t=[]
    for _ in range(n):
        a=time.perf_counter_ns(); f(); t.append((time.perf_counter_ns()-a)/1e3)
    return float(np.median(t))
for kind in ("gaussian", "aniso"):
    corpus = bench.gen_rows(eng, 0, 1294, 1536, 44, "fp32", kind, 1294)
    eng.set_corpus_tensor(corpus)
    q = bench.aniso_queries(eng, 64, 1536, 44) if kind == "aniso" else bench.host_queries(64, 1536, 7)
    for nq in (8, 32, 64):
        for k, ms in ((10, 0.0), (50, 0.85), (50, 0.0)):
            thr = np.float32(_native.f32_threshold(ms))
            t = med(lambda: eng.search_batch(q[:nq], k, thr))
            d = eng.get_option("last_direct")
            eng.profile_enable(True); eng.profile_reset()
            for _ in range(50): eng.search_batch(q[:nq], k, thr)
            kms, n = eng.profile_read(_native.KERNEL_SCAN)
            eng.profile_enable(False)
            print(f"{kind} nq={nq} k={k} ms={ms}: {t:.1f} us (direct {d}), scan kernel {kms/max(n,1)*1e3:.1f} us x {n/50:.0f}", flush=True)
