#!/usr/bin/env python3
"""Small corpora x batches of 2 .. 128 queries, host-synchronous (`tavb_search_batch`: what `VectorBase.fuzzy_lookup_embeddings` and the batched
`lookup_terms` patch call): median us per call the way the library routes it by default, with the grouped one-launch form off
(`direct_group_max_nq` = 0: the plain one-launch form up to 8 queries, the 32/64-query tile and the wide tile beyond -- the routing until the
end of round 6) and with 1 / 2 / 4 / 8 queries per group forced.  Answers of every form are compared with the default's (bit for bit).

    python tools/group_sweep.py [--dtype fp32,fp16] [--dim 1536] [--rows 1000,1294,...] [--sizes 2,4,...] [--k 10,50] [--forms default,off,1,2,4,8]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from typeagent_py_amd import _native  # noqa: E402


def med(f, n=120):
    for _ in range(25):
        f()
    t = np.empty(n)
    for i in range(n):
        a = time.perf_counter_ns()
        f()
        t[i] = (time.perf_counter_ns() - a) / 1e3
    return float(np.median(t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp32,fp16")
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--rows", default="1000,1294,2500,5000,10000,20000,40000")
    ap.add_argument("--sizes", default="2,4,5,8,9,16,24,32,48,64")
    ap.add_argument("--k", default="10,50")
    ap.add_argument("--forms", default="default,off,1,2,4,8")
    ap.add_argument("--min-score", type=float, default=0.0)
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value, set once (e.g. scan_nt=0)")
    ap.add_argument("--kernel", action="store_true", help="(scan kernel us) behind the grouped forms' times (HIP events)")
    args = ap.parse_args()
    sizes = [int(x) for x in args.sizes.split(",")]
    forms = args.forms.split(",")
    eng = _native.Engine(0)
    for o in args.opt:
        name, val = o.split("=")
        eng.set_option(name, int(val))
    qs = bench.host_queries(max(sizes), args.dim, 7)
    thr = np.float32(args.min_score)
    dflt = {n: eng.get_option(n) for n in ("direct_group_max_nq", "direct_group", "direct_group_wgs", "scan_waves")}

    def set_form(form):  # "default" | "off" | "G" | "G@WGS" | "G@WGSwWAVES"
        eng.set_option("direct_group_max_nq", 0 if form == "off" else dflt["direct_group_max_nq"])
        g, _, rest = form.partition("@")
        eng.set_option("direct_group", int(g) if g.isdigit() else dflt["direct_group"])
        wgs, waves = dflt["direct_group_wgs"], dflt["scan_waves"]
        if rest:
            if "w" in rest:
                rest, waves = rest.split("w")[0], int(rest.split("w")[1])
            wgs = int(rest) if rest else wgs
        eng.set_option("direct_group_wgs", wgs)
        eng.set_option("scan_waves", waves)

    for dtype in args.dtype.split(","):
        for k in [int(x) for x in args.k.split(",")]:
            print(f"\n## {dtype}, D = {args.dim}, k = {k}, min_score {args.min_score}: us per `tavb_search_batch` call; forms: {' / '.join(forms)} "
                  f"(off = no grouped form; N = N queries per group; `-` = the form did not take the batch: same route as off)\n")
            print("| rows \\ queries | " + " | ".join(str(n) for n in sizes) + " |\n|---|" + "---|" * len(sizes))
            for rows in [int(x) for x in args.rows.split(",")]:
                corpus = bench.make_device_corpus(eng, rows, args.dim, 50_041, dtype)
                eng.set_corpus_tensor(corpus)
                cells = []
                for nq in sizes:
                    q = qs[:nq]
                    ref = None  # answers of the first streaming form (bit for bit the sequential lookups'); the tiles differ inside float32 noise
                    parts = []
                    for form in forms:
                        set_form(form)
                        o, s, c = eng.search_batch(q, k, thr)
                        took = eng.get_option("last_direct")
                        if form[0].isdigit() and took != 3:
                            parts.append("-")
                            continue
                        valid = np.arange(k)[None, :] < c[:, None]  # (entries past a query's count are not written)
                        o, s = np.where(valid, o, -1), np.where(valid, s, 0).astype(np.float32)
                        bad = ""
                        if took in (1, 3):
                            if ref is None:
                                ref = (o, s, c.copy())
                            elif not (np.array_equal(o, ref[0]) and np.array_equal(s.view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(c, ref[2])):
                                bad = " MISMATCH"
                        us = med(lambda: eng.search_batch(q, k, thr))
                        kern = ""
                        if args.kernel and took == 3:
                            eng.profile_enable(True)
                            eng.profile_reset()
                            for _ in range(40):
                                eng.search_batch(q, k, thr)
                            ms, n = eng.profile_read(_native.KERNEL_SCAN)
                            eng.profile_enable(False)
                            kern = f"({ms / max(n, 1) * 1e3:.0f})"
                        parts.append(f"{us:.0f}{kern}" + (f"[{took}]" if form == "default" else "") + bad)
                    cells.append(" / ".join(parts))
                print(f"| {rows} | " + " | ".join(cells) + " |", flush=True)
            set_form("default")


if __name__ == "__main__":
    main()
