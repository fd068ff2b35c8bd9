#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-parity"
S="--workload cfg3 --rows 1250000 --steps 40 $Q"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccl" > gpurun_out/r3/exp3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/exp3_tests.log
timeout 1500 python tools/bench_variants.py gpurun_out/r3/exp3 \
  "shard_default: $S" \
  "shard_s20480_l4: $S --opt mfma_sample_rows=20480" \
  "shard_s20480_l8: $S --opt mfma_sample_rows=20480 --opt mfma_ladder=8" \
  "shard_s40960_l8: $S --opt mfma_sample_rows=40960 --opt mfma_ladder=8" \
  "shard_s40960_l2: $S --opt mfma_sample_rows=40960 --opt mfma_ladder=2" \
  "shard_s81920_l4: $S --opt mfma_sample_rows=81920 --opt mfma_ladder=4" \
  "shard_tile128: $S --opt mfma_tile=128" \
  "shard2_default: --workload cfg3 --rows 2500000 --steps 30 $Q" \
  "shard4_default: --workload cfg3 --rows 5000000 --steps 20 $Q" \
  "full_l8: --workload cfg3 $Q --opt mfma_ladder=8" \
  "full_s20480: --workload cfg3 $Q --opt mfma_sample_rows=20480" 2>&1 | tee gpurun_out/r3/exp3_variants.txt
