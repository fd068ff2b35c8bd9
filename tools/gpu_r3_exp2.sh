#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-parity"
timeout 1500 python tools/bench_variants.py gpurun_out/r3/exp2 \
  "b32_default: --workload cfg2_b32 $Q" \
  "b32_single: --workload cfg2_b32 $Q --opt mfma_sample_rows=-1" \
  "b32_s65536_l0: --workload cfg2_b32 $Q --opt mfma_sample_rows=65536 --opt mfma_ladder=0" \
  "b32_s131072_l0: --workload cfg2_b32 $Q --opt mfma_sample_rows=131072 --opt mfma_ladder=0" \
  "b32_s262144_l0: --workload cfg2_b32 $Q --opt mfma_sample_rows=262144 --opt mfma_ladder=0" \
  "b32_s131072_l4: --workload cfg2_b32 $Q --opt mfma_sample_rows=131072 --opt mfma_ladder=4" \
  "b8_default: --workload cfg2_b32 --queries 8 $Q" \
  "b16_default: --workload cfg2_b32 --queries 16 $Q" \
  "b64_default: --workload cfg2_b32 --queries 64 $Q" \
  "b64_s131072_l0: --workload cfg2_b32 --queries 64 $Q --opt mfma_sample_rows=131072 --opt mfma_ladder=0" \
  "f16_b32_default: --workload cfg3_b32 --rows 2000000 $Q" \
  "f16_b32_s131072_l0: --workload cfg3_b32 --rows 2000000 $Q --opt mfma_sample_rows=131072 --opt mfma_ladder=0" \
  "f16_b32_10M_default: --workload cfg3_b32 $Q" \
  "f16_b32_10M_s131072_l4: --workload cfg3_b32 $Q --opt mfma_sample_rows=131072 --opt mfma_ladder=4" 2>&1 | tee gpurun_out/r3/exp2_variants.txt
