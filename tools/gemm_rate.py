#!/usr/bin/env python3
"""What the vendor GEMM (hipBLASLt through torch.matmul) reaches on the cfg3 contraction shape on this box, next to the fused
kernel: [rows, 1536] fp16 x [1536, 1024] fp16 -> fp16, fp32 accumulate, NO top-k, the [rows, 1024] product written out.
Measurement tool only (profiles/r02_cfg3_power.md); nothing in the product uses torch.matmul."""
import json
import sys

import torch


def rate(rows: int, nq: int, dim: int, zeros: bool, iters: int) -> float:
    a = torch.zeros((rows, dim), dtype=torch.float16, device="cuda") if zeros else torch.randn((rows, dim), dtype=torch.float16, device="cuda") * 0.0255
    b = torch.zeros((dim, nq), dtype=torch.float16, device="cuda") if zeros else torch.randn((dim, nq), dtype=torch.float16, device="cuda") * 0.0255
    out = torch.empty((rows, nq), dtype=torch.float16, device="cuda")
    for _ in range(3):
        torch.matmul(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * rows * nq * dim / (ms * 1e-3) / 1e12


def main() -> None:
    res = {}
    for rows in (327_680, 1_310_720):
        for zeros in (False, True):
            res[f"rows{rows}_{'zeros' if zeros else 'gaussian'}"] = round(rate(rows, 1024, 1536, zeros, 60 if rows < 1_000_000 else 20), 1)
    res["unit"] = "TFLOP/s; torch.matmul fp16 [rows,1536]x[1536,1024], output written, 60/20 back-to-back calls"
    json.dump(res, sys.stdout)
    print()


if __name__ == "__main__":
    main()
