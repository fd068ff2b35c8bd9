#!/bin/bash
# round 2, probe 1: operand-delivery micro-benchmarks (full-line pieces) + baseline cfg3 numbers on this box
mkdir -p gpurun_out/r2p1
./tools/microbench/issue_cost > gpurun_out/r2p1/issue_cost.txt 2>&1
./tools/microbench/load_paths > gpurun_out/r2p1/load_paths.txt 2>&1
for v in "" "--opt mfma_variant=5" "--opt mfma_variant=5 --opt mfma_ablate=256" "--opt mfma_variant=5 --opt mfma_ablate=512"; do
  python bench.py --workload cfg3 --no-cpu-baseline --steps 5 --warmup 2 $v >> gpurun_out/r2p1/cfg3.jsonl 2>> gpurun_out/r2p1/cfg3.err
done
tail -n 40 gpurun_out/r2p1/issue_cost.txt
