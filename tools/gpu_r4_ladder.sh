#!/bin/bash
# round 4: smaller first phases on a 1/8 shard (the all-admitted first phase is where the admission cost sits: 0.36 ms of 3.45)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/ladder; mkdir -p $O
Q="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --rows 1250000 --steps 40 --warmup 10"
timeout 1500 python tools/bench_variants.py $O \
  "s20480x4: $Q" \
  "s2560x4: $Q --opt mfma_sample_rows=2560" \
  "s5120x4: $Q --opt mfma_sample_rows=5120" \
  "s10240x4: $Q --opt mfma_sample_rows=10240" \
  "s2560x8: $Q --opt mfma_sample_rows=2560 --opt mfma_ladder=8" \
  "s5120x8: $Q --opt mfma_sample_rows=5120 --opt mfma_ladder=8" \
  "s2560x3: $Q --opt mfma_sample_rows=2560 --opt mfma_ladder=3" \
  "s20480x4_again: $Q" \
  "full_s2560x4: --no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 20 --warmup 5 --opt mfma_sample_rows=2560" \
  "full_s5120x8: --no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 20 --warmup 5 --opt mfma_sample_rows=5120 --opt mfma_ladder=8" \
  "full: --no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 20 --warmup 5" 2>&1 | tee $O/variants.txt
