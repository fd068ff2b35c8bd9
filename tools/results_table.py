#!/usr/bin/env python3
"""Markdown table of one bench.py line (headline + sub-records): workload, step time, value, roofline fraction, parity.

    python tools/results_table.py profiles/r05_bench_default.json
"""
import json
import sys


def row(name, r, headline=False):
    ro, pa = r.get("roofline") or {}, r.get("parity") or {}
    val = r.get("value", r.get("queries_per_sec", 0.0))
    unit = r.get("unit", "queries/s")
    wl = (r.get("config") or {}).get("workload") or r.get("workload", "")
    wl = wl.split(": ", 1)[-1]
    par = "ok" if pa.get("ok") else ("FAILED" if pa else "–")
    if pa.get("positions_exact") is not None:
        par += f" ({pa['positions_exact']} exact" + (f" + {pa['positions_permuted']} permuted inside float32 ties" if pa.get("positions_permuted") else "") + ")"
    extra = []
    if r.get("flagged_fraction"):
        extra.append(f"flagged {r['flagged_fraction']:g}")
    if "vs_gaussian" in r:
        extra.append(f"{r['vs_gaussian']:.2f}x the gaussian rate")
    if ro.get("traffic"):
        alg = ro.get("algorithmic_per_step")
        extra.append(f"traffic {ro['traffic'] / 1e9:.1f} GB")
    frac = f"**{ro.get('frac', 0):.3f}** {ro.get('bound', '')}" if ro else "–"
    if ro.get("bound") == "latency":
        frac = f"launch-bound (kernel {ro.get('kernel_us_per_step', 0):.1f} µs)"
    if ro.get("frac_at_clock"):
        frac += f" ({ro['frac_at_clock']:.2f} at {ro.get('sclk_mhz', 0):.0f} MHz)"
    ca = r.get("class_api") or {}
    if not headline and ca.get("scored_int_lists"):
        extra.append(f"class {ca['scored_int_lists']['ms_per_step'] * 1e3:.1f} µs" if ca['scored_int_lists']['ms_per_step'] < 0.2 else f"class {ca['scored_int_lists']['ms_per_step']:.2f} ms")
    if ca.get("fresh_list_every_call"):
        f_ms = ca["fresh_list_every_call"]["ms_per_step"]
        extra.append(f"a fresh list every call {f_ms * 1e3:.0f} µs" if f_ms < 0.2 else f"a fresh list every call {f_ms:.1f} ms")
    cb = r.get("cpu_baseline") or {}
    if not headline and cb.get("p50_ms_per_query_on_sample"):
        c_ms = cb["p50_ms_per_query_on_sample"]
        extra.append(f"CPU {cb.get('kind', '')} {c_ms * 1e3:.0f} µs" if c_ms < 0.2 else f"CPU {cb.get('kind', '')} {c_ms:.1f} ms per query")
    ms = r.get("ms_per_step", 0.0)
    ms_s = f"{ms * 1e3:.1f} µs" if ms < 0.2 else f"{ms:.2f} ms"
    v = f"{val / 1e3:.1f} k" if val >= 1e4 else f"{val:.1f}"
    return f"| {'**' + name + '**' if headline else name} | {wl} | {ms_s} | {v} {unit} | {frac} | {par}{'; ' + ', '.join(extra) if extra else ''} |"


def main():
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("| record | workload | per step | value | roofline `frac` | parity (whole corpus) |")
    print("|---|---|---|---|---|---|")
    print(row("cfg3 (headline)", d, True))
    for name, r in (d.get("sub") or {}).items():
        print(row(name, r))
        variants = r.get("variants") or {}
        if variants and not any(isinstance(v, dict) for v in variants.values()):  # cfg1_terms32: the same call three ways, us per call
            print(f"| {name}.variants | `tavb_search_batch` (host queries in, host results out) | | | – | as ONE grouped launch {variants.get('grouped_us', 0):.0f} µs, "
                  f"on the tiles (grouped form off) {variants.get('tiles_us', 0):.0f} µs, as sequential single lookups {variants.get('sequential_us', 0):.0f} µs |")
            continue
        for vn, v in variants.items():
            print(f"| {name}.{vn} | | {v['ms_per_step']:.2f} ms | {v['value']:.1f} user-queries/s | " + (f"{v['hbm_frac']:.3f} hbm" if "hbm_frac" in v else "–") + " | " + ("ok" if (v.get("parity") or {}).get("ok") else "–")
                  + (f"; fused is {v['fused_speedup']:.2f}x these six separate calls" if "fused_speedup" in v else "") + " |")
    ca = d.get("class_api") or {}
    if ca:
        print(f"\nThrough the class (`VectorBase.fuzzy_lookup_embeddings`, host queries in, Python objects out): `list[list[ScoredInt]]` "
              f"{ca['scored_int_lists']['ms_per_step']:.2f} ms per batch, `as_arrays=True` {ca['as_arrays']['ms_per_step']:.2f} ms; engine {d['ms_per_step']:.2f} ms; "
              f"host-buffer form of the C call {d['host_buffer_form']['ms_per_step']:.2f} ms.")
    cb = d.get("cpu_baseline") or {}
    if cb:
        print(f"CPU baseline beside the headline ({cb['kind']}, {cb['host_cores']} host cores): {cb['value']:.2f} queries/s at the best thread count ({cb['cores']}), "
              f"{cb['default_threads']['value']:.2f} at OpenBLAS's default, {cb.get('one_thread', {}).get('value', 0):.2f} on one thread.")
    su = (d.get("roofline") or {}).get("sustained")
    if su:
        print(f"`roofline.sustained`: MFMA-only {su['mfma_only_tflops']:.0f} TFLOP/s, vendor GEMM {su['vendor_gemm_tflops']:.0f} TFLOP/s, shipping / MFMA-only {su['frac_of_mfma_only']:.2f}.")


if __name__ == "__main__":
    main()
