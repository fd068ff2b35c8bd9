#!/bin/bash
# round 4: mid-batch records after the ladder / admission changes (first phase 10240 vs 20480 rows for the 128-query tile), cfg2 on this box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/mid; mkdir -p $O
Q="--no-cpu-baseline --no-parity --no-sub --no-calibration"
timeout 1700 python tools/bench_variants.py $O \
  "b128: $Q --workload cfg3_b128" "b128_s20480: $Q --workload cfg3_b128 --opt mfma_sample_rows=20480" "b128_s40960: $Q --workload cfg3_b128 --opt mfma_sample_rows=40960" "b128_again: $Q --workload cfg3_b128" \
  "q256: $Q --workload cfg3 --queries 256" "q256_s20480: $Q --workload cfg3 --queries 256 --opt mfma_sample_rows=20480" \
  "q512: $Q --workload cfg3 --queries 512" "q512_s20480: $Q --workload cfg3 --queries 512 --opt mfma_sample_rows=20480" \
  "cfg2: $Q --workload cfg2" "cfg2_again: $Q --workload cfg2" "cfg2_f16: $Q --workload cfg2_f16" "cfg3_q1: $Q --workload cfg3_q1" 2>&1 | tee $O/variants.txt
