#!/usr/bin/env python3
"""Does the power-limited clock of the cfg3 contraction depend on how many mantissa bits of the QUERY operand are non-zero?

The board sits at its power cap under dense MFMA work and hipBLASLt runs 1.3x faster on zeros than on gaussian operands
(profiles/r06_raw/gemm_rate_zeros_vs_gaussian.txt).  The wide tile is a FILTER (tavb_rescore.hip): it may multiply any rounding of the
queries as long as delta_q bounds the rounding rigorously.  This script measures what a coarser query rounding would buy before anything is
built: the shipping lookup (and its MFMA-only ablation) on queries whose fp16 mantissa keeps only its top m bits (the values handed over are
exactly representable, so the library's own fp16 rounding is the identity), and the same with the CORPUS truncated (for the other operand's
sensitivity; not something the product could do).  Board power and clock come from the same loops (tools/ceiling.py's sampler).

    python tools/mantissa_power.py [--seconds 5] [--rows 10000000] [--bits 10,8,7,6,5,4,2,0]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from ceiling import leg  # noqa: E402


def truncate_f16(t, m):
    """fp16 tensor -> the same with only the top m of the 10 stored mantissa bits kept (round to nearest on the dropped bits)."""
    import torch

    if m >= 10:
        return t.clone()
    drop = 10 - m
    bits = t.view(torch.int16).to(torch.int32) & 0xFFFF
    half = 1 << (drop - 1)
    mag = (bits & 0x7FFF) + half
    mag = mag & ~((1 << drop) - 1)
    mag = torch.clamp(mag, max=0x7BFF)
    out = ((bits & 0x8000) | mag).to(torch.int32)
    out = torch.where(out >= 0x8000, out - 0x10000, out).to(torch.int16)
    return out.view(torch.float16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--bits", default="10,8,7,6,5,4,2,0")
    ap.add_argument("--corpus-bits", default="6,0")
    args = ap.parse_args()
    import torch

    from typeagent_py_amd import _native

    eng = _native.Engine(0)
    wl = bench.WORKLOADS["cfg3"]
    nq, k, dim = wl["nq"], wl["k"], wl["dim"]
    corpus = bench.gen_rows(eng, 0, args.rows, dim, wl["seed"], "fp16")
    eng.set_corpus_tensor(corpus)
    q16 = torch.from_numpy(bench.host_queries(nq, dim, 4242)).to("cuda:0").to(torch.float16)
    keys = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)
    flops = 2.0 * nq * args.rows * dim
    print(f"rows {args.rows}, {nq} queries, D {dim}, fp16; {args.seconds:g} s per leg")
    print("| operand truncated | mantissa bits kept | leg | TFLOP/s | ms per call | power W | sclk MHz |")
    print("|---|---|---|---|---|---|---|")

    def run(label, m, dq):
        def lookup():
            eng.search_device(dq, k, 0.0, out_keys=keys)

        for name, abl in (("shipping", 0), ("mfma_only", 258)):
            eng.set_option("mfma_ablate", abl)
            r = leg(lookup, args.seconds, flops)
            print(f"| {label} | {m} | {name} | {r['tflops']:.0f} | {r['ms_per_call']:.2f} | {r['power_w']:.0f} | {r['sclk_mhz']:.0f} |", flush=True)
        eng.set_option("mfma_ablate", 0)

    for m in [int(x) for x in args.bits.split(",") if x != ""]:
        run("queries", m, truncate_f16(q16, m).to(torch.float32).contiguous())
    full = corpus.clone()
    for m in [int(x) for x in args.corpus_bits.split(",") if x != ""]:
        for lo in range(0, args.rows, 1_000_000):  # in chunks: the int32 temporaries of a 30 GB corpus would not fit
            corpus[lo:lo + 1_000_000].copy_(truncate_f16(full[lo:lo + 1_000_000], m))
        eng.set_corpus_tensor(corpus)
        run("corpus", m, q16.to(torch.float32).contiguous())
        run("corpus+queries", m, truncate_f16(q16, m).to(torch.float32).contiguous())


if __name__ == "__main__":
    main()
