#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc ... --kernel-trace --output-format csv` pass: per kernel the dispatch count, the mean duration
and the per-dispatch mean of every counter; with --steps N also the FETCH_SIZE / WRITE_SIZE totals per bench step over the
lookup kernels (corpus generation -- torch RNG, normalize_rows, f32_to_f16 -- excluded).

    python tools/pmc_summary.py gpurun_out/x/*/*_counter_collection.csv --steps 3 [--json out.json --name cfg3]

FETCH_SIZE counts KiB; on gfx950 a wide coalesced 16 B/lane stream is counted at half its bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so HBM bytes read = FETCH_SIZE x 1024 x 2 for these kernels.
"""
import argparse
import collections
import csv
import json
import re
import sys

SETUP = ("at::native", "rocprim::", "normalize_rows_kernel", "f32_to_f16_kernel", "corpus_max_norm_kernel", "shadow_convert_kernel", "__amd_rocclr_copyBuffer")


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(tavb::\w+(?: const)?[&*]?(?:, [^)]*)?\)$", "", name)
    name = name.replace("void ", "").replace("tavb::", "")
    return name if len(name) <= 100 else name[:97] + "..."


def kernel_base(name: str) -> str | None:
    """`scan_fixed_kernel` out of a demangled or mangled libtavb kernel name; None for kernels that are not ours (runtime fills, torch).
    (rocprofv3 leaves names with _Float16 template arguments mangled, and binutils' c++filt does not know `DF16_` either: the nested name
    is read off the Itanium length prefixes.)"""
    if name.startswith("_ZN"):
        rest, comps = name[3:], []
        while rest and rest[0].isdigit():
            digits = re.match(r"\d+", rest).group(0)
            comps.append(rest[len(digits) : len(digits) + int(digits)])
            rest = rest[len(digits) + int(digits) :]
        return comps[-1] if comps and comps[0] == "tavb" and comps[-1].endswith("_kernel") else None
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    m = re.match(r"(?:tavb::)?([a-z][a-z0-9_]*_kernel)\b", name)
    return m.group(1) if m else None


def is_split(name: str) -> bool:
    """The SPLIT instantiation of the tile kernel: `mfma_scan_kernel<ABL, NI, N3, N0, N1, SPLIT = true, BD>`."""
    return re.search(r"mfma_scan_kernel<[^>]*, true, (?:true|false)>", name) is not None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("csv", nargs="+")
    ap.add_argument("--steps", type=int, default=0, help="lookups the profiled run made (bench.py prints the number on stderr)")
    ap.add_argument("--cmd", default=None, help="the profiled command (for the header)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--name", default=None)
    ap.add_argument("--round", default="r05", help="prefix of the profiles/ file names this summary is committed under")
    ap.add_argument("--script", default="tools/gpu.sh pmc")
    args = ap.parse_args()
    csv.field_size_limit(1 << 30)
    per = collections.defaultdict(lambda: collections.defaultdict(float))  # kernel -> counter -> sum
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for path in args.csv:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                d = (path, row["Dispatch_Id"])
                per[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k].add(d)
                if d not in seen:
                    seen.add(d)
                    dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    counters = sorted({c for k in per for c in per[k]})
    if args.cmd:
        print(f"# rocprofv3 --pmc pass (MI355X): `cd /tmp && rocprofv3 --pmc {' '.join(counters)} --kernel-trace --output-format csv -- python {args.cmd}`\n")
        print("Per-dispatch means (tools/pmc_summary.py).  SQ_* wave counters count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per 32x32x16 MFMA, "
              "summed over all SIMDs), GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE are KiB.\n")
    print("| kernel | dispatches | mean us | " + " | ".join(counters) + " |")
    print("|---|---|---|" + "---|" * len(counters))
    for k in sorted(per, key=lambda k: -dur[k]):
        n = len(disp[k])
        print(f"| `{k}` | {n} | {dur[k] / n / 1e3:.1f} | " + " | ".join(f"{per[k].get(c, 0.0) / n:.6g}" for c in counters) + " |")
    if args.steps:
        out = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c in counters:
                tot = sum(per[k][c] for k in per if not any(s in k for s in SETUP))
                out[c] = tot / args.steps
        if "FETCH_SIZE" in out:
            b = out["FETCH_SIZE"] * 1024 * 2
            print(f"\nThe run made {args.steps} lookups.  Lookup kernels (corpus generation and the one-time per-corpus kernels excluded), per lookup: "
                  f"FETCH_SIZE {out['FETCH_SIZE']:.6g} KiB x 1024 x 2 = **{b / 1e9:.3f} GB read**"
                  + (f", WRITE_SIZE {out['WRITE_SIZE']:.6g} KiB" if "WRITE_SIZE" in out else ""))
            if args.json and args.name:
                try:
                    blob = json.load(open(args.json))
                except Exception:
                    blob = {}
                lookup_kernels = sorted({kernel_base(k) for k in per if not any(s_ in k for s_ in SETUP) and per[k].get("FETCH_SIZE", 0.0) > 0} - {None})
                blob[args.name] = {
                    "traffic_bytes_per_step": b,
                    "counter": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950 half-count correction, MI355X_MICROARCH.md HBM section), summed over every lookup kernel of a step",
                    "source": f"rocprofv3 --pmc FETCH_SIZE --kernel-trace pass of `{args.cmd or 'bench.py'}` ({args.steps} lookups in the run), {args.script} + tools/pmc_summary.py, profiles/{args.round}_pmc_{args.name}_fetch.md",
                    # the kernels the sum runs over: tests/test_bench_contract.py checks that each still exists in libtavb.so (a stale file must not go unnoticed)
                    "kernels": lookup_kernels,
                    # launches of each lookup kernel per lookup (the tile kernel: one per ladder phase -- checked against tavb_plan_ladder() by the same test)
                    # (the SPLIT instantiation of the tile kernel -- the exact fallback, launched per ladder phase and returning at once when no query is
                    #  flagged -- is counted apart: `..., true, false>` in the demangled name)
                    "launches_per_step": {kb + suffix: round(sum(len(disp[k]) for k in per if kernel_base(k) == kb and is_split(k) == (suffix != "")) / args.steps, 4)
                                          for kb in lookup_kernels for suffix in ("", "_split") if any(kernel_base(k) == kb and is_split(k) == (suffix != "") for k in per)},
                }
                json.dump(blob, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
