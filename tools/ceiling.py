#!/usr/bin/env python3
"""What this box sustains on the cfg3 contraction (10M x 1536 fp16 rows x 1024 queries), with board power and clock from the SAME loops:

    shipping     the lookup as it ships (tile kernel with staging, admissions, selection, rescoring)
    mfma_only    the tile kernel with operand staging and admissions compiled out (option mfma_ablate=258): MFMA stream, fragment reads, barriers
    vendor_gemm  torch.matmul (hipBLASLt) on [327680, 1536] rows of the corpus x the queries, product written, no selection

Each leg runs for about `--seconds` while a sampler thread reads `rocm-smi --showpower --showclocks` every 0.4 s; the table printed at
the end (markdown) is DESIGN.md's "practical ceiling".  Usage on the GPU box:  python tools/ceiling.py [--seconds 8] [--rows 10000000]
"""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = threading.Event()
        self.power, self.sclk = [], []

    def run(self):
        while not self.stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                out = ""
            for line in out.splitlines():
                if "GPU[0]" not in line:
                    continue
                m = re.search(r"Package Power \(W\):\s*([0-9.]+)", line)
                if m:
                    self.power.append(float(m.group(1)))
                m = re.search(r"sclk clock level:.*\((\d+)Mhz\)", line)
                if m:
                    self.sclk.append(float(m.group(1)))
            self.stop.wait(0.4)


def leg(fn, seconds, flops_per_call):
    import torch

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    time.sleep(0.2)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    s.stop.set()
    s.join()
    med = lambda a: float(np.median(a[len(a) // 4:])) if a else float("nan")  # (the first quarter: the clocks are still settling)
    return {"tflops": flops_per_call * n / el / 1e12, "ms_per_call": el / n * 1e3, "power_w": med(s.power), "sclk_mhz": med(s.sclk), "samples": len(s.power)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--rows", type=int, default=10_000_000)
    args = ap.parse_args()
    import torch

    from typeagent_py_amd import _native

    eng = _native.Engine(0)
    wl = bench.WORKLOADS["cfg3"]
    nq, k, dim = wl["nq"], wl["k"], wl["dim"]
    corpus = bench.gen_rows(eng, 0, args.rows, dim, wl["seed"], "fp16")
    eng.set_corpus_tensor(corpus)
    dq = torch.from_numpy(bench.host_queries(nq, dim, 4242)).to("cuda:0")
    keys = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)
    flops = 2.0 * nq * args.rows * dim

    def lookup():
        eng.search_device(dq, k, 0.0, out_keys=keys)

    out = {"shipping": leg(lookup, args.seconds, flops)}
    eng.set_option("mfma_ablate", 258)
    out["mfma_only"] = leg(lookup, args.seconds, flops)
    eng.set_option("mfma_ablate", 0)
    g_rows = min(args.rows, 327_680)
    a, b = corpus[:g_rows], dq.to(torch.float16)
    prod = torch.empty((g_rows, nq), dtype=torch.float16, device=a.device)
    out["vendor_gemm"] = leg(lambda: torch.matmul(a, b.t(), out=prod), args.seconds, 2.0 * g_rows * nq * dim)
    cap = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout
    m = re.search(r"GPU\[0\].*?([0-9.]+)\s*$", cap, re.M)
    print(f"power cap: {m.group(1) if m else '?'} W; rows {args.rows}, {nq} queries, D {dim}, fp16; {args.seconds:g} s per leg")
    print("| leg | TFLOP/s | of 2500 | ms per call | board power (median W) | sclk (median MHz) |")
    print("|---|---|---|---|---|---|")
    for name, r in out.items():
        print(f"| {name} | {r['tflops']:.0f} | {r['tflops'] / 2500:.3f} | {r['ms_per_call']:.2f} | {r['power_w']:.0f} | {r['sclk_mhz']:.0f} |")


if __name__ == "__main__":
    main()
