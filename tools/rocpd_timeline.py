#!/usr/bin/env python3
"""One lookup's launches in order, from a rocprofv3 rocpd database (kernel trace): start offset, duration and the idle gap in front of
every kernel, plus totals (busy / idle / launches per lookup) over the steady-state lookups.  A "lookup" is delimited by the kernel named
`--first` (default: the first launch of the wide path, query_prepare_kernel).

    python tools/rocpd_timeline.py x_results.db [--first query_prepare] [--skip 5] [--show 1]
"""
import argparse
import re
import sqlite3


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("tavb::", "")
    m = re.match(r"_ZN4tavb(?:12_GLOBAL__N_1)?(\d+)", name)
    if m:
        n = int(m.group(1))
        name = name[m.end():m.end() + n]
    return name.split("(")[0][:60]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--first", default="query_prepare")
    ap.add_argument("--skip", type=int, default=5, help="lookups to skip (warm-up)")
    ap.add_argument("--show", type=int, default=1, help="lookups to print launch by launch")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    groups, cur_g = [], None
    for name, s, e in rows:
        sn = short(name)
        if args.first in sn:
            cur_g = []
            groups.append(cur_g)
        if cur_g is not None:
            cur_g.append((sn, s, e))
    groups = [g for g in groups[args.skip:] if g]
    if len(groups) > 1:
        groups = groups[:-1]  # (the last one may carry the teardown)
    print(f"# {args.db}: {len(groups)} lookups after skipping {args.skip}; delimiter kernel `{args.first}`\n")
    tot_busy = tot_span = tot_n = 0
    for g in groups:
        tot_busy += sum(e - s for _, s, e in g)
        tot_span += g[-1][2] - g[0][1]
        tot_n += len(g)
    n = max(len(groups), 1)
    print(f"per lookup: {tot_n / n:.1f} launches, span {tot_span / n / 1e3:.1f} us first start -> last end, kernels busy {tot_busy / n / 1e3:.1f} us, "
          f"idle between them {(tot_span - tot_busy) / n / 1e3:.1f} us\n")
    for g in groups[:args.show]:
        print("| # | kernel | start us | duration us | idle gap in front us |")
        print("|---|---|---|---|---|")
        t0, prev_end = g[0][1], g[0][1]
        for i, (sn, s, e) in enumerate(g):
            print(f"| {i} | `{sn}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} |")
            prev_end = max(prev_end, e)
        print()


if __name__ == "__main__":
    main()
