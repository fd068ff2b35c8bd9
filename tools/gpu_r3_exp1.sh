#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "captured_graph or rccl" > gpurun_out/r3/exp1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/exp1_tests.log
timeout 300 python tools/latency_breakdown.py 10000 2>&1 | tee gpurun_out/r3/latency_cfg1.txt
Q="--no-cpu-baseline --no-parity"
timeout 1500 python tools/bench_variants.py gpurun_out/r3/exp1 \
  "cfg1: --workload cfg1 $Q" \
  "shard_default: --workload cfg3 --rows 1250000 --steps 30 $Q" \
  "shard_ladder0: --workload cfg3 --rows 1250000 --steps 30 $Q --opt mfma_ladder=0" \
  "shard_ladder16: --workload cfg3 --rows 1250000 --steps 30 $Q --opt mfma_ladder=16" \
  "shard_single: --workload cfg3 --rows 1250000 --steps 30 $Q --opt mfma_sample_rows=-1" \
  "shard_sample81920: --workload cfg3 --rows 1250000 --steps 30 $Q --opt mfma_sample_rows=81920 --opt mfma_ladder=0" \
  "one_rank_dist: TAVB_BENCH_FORCE_DIST=1 --workload cfg3 --rows 1250000 --steps 30 --no-cpu-baseline" \
  "cfg2_b32: --workload cfg2_b32 $Q" \
  "cfg2_b1024: --workload cfg2_b1024 $Q" \
  "cfg3_b128: --workload cfg3_b128 $Q" 2>&1 | tee gpurun_out/r3/exp1_variants.txt
