#!/bin/bash
# round 4: cfg2_b32 (32 queries on the fp32 32-query tile, 1M rows): size of the seeding phase
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/b32; mkdir -p $O
Q="--no-cpu-baseline --no-parity --workload cfg2_b32"
timeout 1700 python tools/bench_variants.py $O "auto: $Q" "s32768: $Q --opt mfma_sample_rows=32768 --opt mfma_ladder=0" "s65536: $Q --opt mfma_sample_rows=65536 --opt mfma_ladder=0" \
  "s16384: $Q --opt mfma_sample_rows=16384 --opt mfma_ladder=0" "s65536x4: $Q --opt mfma_sample_rows=65536" "s16384x8: $Q --opt mfma_sample_rows=16384 --opt mfma_ladder=8" "auto_again: $Q" \
  "b8: $Q --queries 8" "b16: $Q --queries 16" "b64: $Q --queries 64" 2>&1 | tee $O/variants.txt
