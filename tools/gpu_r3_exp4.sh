#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-parity"
timeout 1500 python tools/bench_variants.py gpurun_out/r3/exp4 \
  "b32_default: --workload cfg2_b32 $Q --steps 50" \
  "b32_s8192_l0: --workload cfg2_b32 $Q --steps 50 --opt mfma_sample_rows=8192 --opt mfma_ladder=0" \
  "b32_s16384_l0: --workload cfg2_b32 $Q --steps 50 --opt mfma_sample_rows=16384 --opt mfma_ladder=0" \
  "b32_s32768_l0: --workload cfg2_b32 $Q --steps 50 --opt mfma_sample_rows=32768 --opt mfma_ladder=0" \
  "b32_s65536_l0: --workload cfg2_b32 $Q --steps 50 --opt mfma_sample_rows=65536 --opt mfma_ladder=0" \
  "b32_s16384_l8: --workload cfg2_b32 $Q --steps 50 --opt mfma_sample_rows=16384 --opt mfma_ladder=8" \
  "b8_s16384_l0: --workload cfg2_b32 --queries 8 $Q --steps 50 --opt mfma_sample_rows=16384 --opt mfma_ladder=0" \
  "f16_2M_b32_s32768_l0: --workload cfg3_b32 --rows 2000000 $Q --steps 50 --opt mfma_sample_rows=32768 --opt mfma_ladder=0" \
  "f16_2M_b32_default: --workload cfg3_b32 --rows 2000000 $Q --steps 50" 2>&1 | tee gpurun_out/r3/exp4_variants.txt
