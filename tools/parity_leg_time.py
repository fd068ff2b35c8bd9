#!/usr/bin/env python3
"""How long rank 0's whole-corpus parity pass of `bench.py --gpus 8` (sub.cfg4_weak: 100M rows) takes: the other ranks wait for it in a barrier
whose limit is the process group's 45-minute timeout (bench.py Ctx).  Runs that pass as rank 0 would -- its own 12.5M rows resident, the other
87.5M regenerated on the device chunk by chunk, every chunk scored by the CPU oracle for 16 queries with the float64 referee -- on one GPU.
    python tools/parity_leg_time.py [--world 8] [--rows-per-rank 12500000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rows-per-rank", type=int, default=12_500_000)
    args = ap.parse_args()
    from oracle import vectorbase_oracle as vo
    from typeagent_py_amd import _native

    eng = _native.Engine(0)
    wl = dict(bench.WORKLOADS["cfg4"])
    total = args.rows_per_rank * args.world
    t0 = time.perf_counter()
    resident = bench.gen_rows(eng, 0, args.rows_per_rank, wl["dim"], wl["seed"], wl["dtype"], "gaussian", total)
    t_gen = time.perf_counter() - t0
    queries = bench.host_queries(64, wl["dim"], 4242)
    sample = list(range(bench.PARITY_QUERIES))
    returned = [np.arange(32, dtype=np.int64) for _ in sample]  # (stand-ins for the device's answers: the pass costs the same)
    t0 = time.perf_counter()
    n_chunks = [0]

    def chunks():
        for c in bench.oracle_chunks(eng, resident, 0, total, wl["dim"], wl["seed"], wl["dtype"]):
            n_chunks[0] += 1
            if n_chunks[0] % 10 == 0:
                print(f"  {n_chunks[0]} chunks, {time.perf_counter() - t0:.0f} s", flush=True)
            yield c
    ref, referee = vo.scores_full_chunked_refereed(chunks(), queries[sample], returned, keep=wl["k"] + 256)
    t_pass = time.perf_counter() - t0
    t0 = time.perf_counter()
    for j in sample:
        top = np.argpartition(-ref[j], 32 + 256)[: 32 + 256]
        referee.for_query(j)(top)
    t_check = time.perf_counter() - t0
    print(f"rows {total} ({args.world} x {args.rows_per_rank}), {len(sample)} queries: own shard generated in {t_gen:.1f} s; oracle pass over {n_chunks[0]} chunks "
          f"{t_pass:.1f} s ({t_pass / n_chunks[0]:.2f} s per 1M-row chunk); top-k selection + referee lookups {t_check:.1f} s; "
          f"limit of the barrier the other ranks wait in: 2700 s")


if __name__ == "__main__":
    main()
