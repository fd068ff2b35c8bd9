#!/bin/bash
# NOTE: kept as run (mid round 2).  The `mfma_variant` option no longer exists -- variant 6 is the only 256-query kernel --, so drop it to re-run;
# variant 5 (round 1's 4-wave 384 x 256 tile) was deleted after this measurement.
# round 2, probe 1: operand-delivery micro-benchmarks (full-line pieces) + baseline cfg3 numbers on this box
mkdir -p gpurun_out/r2p1
./tools/microbench/issue_cost > gpurun_out/r2p1/issue_cost.txt 2>&1
./tools/microbench/load_paths > gpurun_out/r2p1/load_paths.txt 2>&1
for v in "" "--opt mfma_variant=5" "--opt mfma_variant=5 --opt mfma_ablate=256" "--opt mfma_variant=5 --opt mfma_ablate=512"; do
  python bench.py --workload cfg3 --no-cpu-baseline --steps 5 --warmup 2 $v >> gpurun_out/r2p1/cfg3.jsonl 2>> gpurun_out/r2p1/cfg3.err
done
tail -n 40 gpurun_out/r2p1/issue_cost.txt
