#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output) the way
`--stats` prints it: per kernel calls / total / average / min / max duration and share,
plus per-kernel register/LDS info, and PMC counter sums when the db holds counters.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.md
"""
import sqlite3
import sys


def main(path: str) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, (end - start) as dur, vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x "
                       "from kernels" if {"vgpr_count", "lds_size", "workgroup_x"} <= set(cols) else "select name, (end - start) as dur, 0,0,0,0,0,0 from kernels").fetchall()
    agg = {}
    for name, dur, vg, sg, lds, scr, wg, grid in rows:
        a = agg.setdefault(name, dict(n=0, tot=0, mn=1 << 62, mx=0, vgpr=vg, sgpr=sg, lds=lds, scratch=scr, wg=wg, grid=set()))
        a["n"] += 1
        a["tot"] += dur
        a["mn"] = min(a["mn"], dur)
        a["mx"] = max(a["mx"], dur)
        a["grid"].add(grid)
    total = sum(a["tot"] for a in agg.values()) or 1
    print(f"# rocprofv3 kernel trace summary: {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | wg | grid |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        short = name if len(name) < 110 else name[:107] + "..."
        short = short.replace("|", "\\|")
        grids = ",".join(str(g) for g in sorted(a["grid"])[:4])
        print(f"| `{short}` | {a['n']} | {a['tot'] / 1e6:.3f} | {a['tot'] / a['n'] / 1e3:.2f} | {a['mn'] / 1e3:.2f} | {a['mx'] / 1e3:.2f} | "
              f"{100 * a['tot'] / total:.1f} | {a['vgpr']} | {a['sgpr']} | {a['lds']} | {a['scratch']} | {a['wg']} | {grids} |")
    # counters, if any
    try:
        pm = cur.execute("select k.name, p.counter_name, sum(p.value), count(*) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
                         "group by k.name, p.counter_name").fetchall()
    except Exception:
        try:
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            pm = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall() if ccols else []
        except Exception:
            pm = []
    if pm:
        print("\n| kernel | counter | sum | dispatches | per dispatch |")
        print("|---|---|---|---|---|")
        for name, cname, val, n in pm:
            short = name if len(name) < 90 else name[:87] + "..."
            print(f"| `{short}` | {cname} | {val:.6g} | {n} | {val / max(n, 1):.6g} |")


if __name__ == "__main__":
    main(sys.argv[1])
