#!/bin/bash
# round 4: counted L2 touch-ahead of the corpus slabs (mfma_sched = 3 / 4 / 5: 1 / 2 / 3 steps ahead) against the shipping schedule, one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/touch; mkdir -p $O
Q="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 20 --warmup 5"
timeout 1500 python tools/bench_variants.py $O \
  "ship: $Q" \
  "touch1: $Q --opt mfma_sched=3" \
  "touch2: $Q --opt mfma_sched=4" \
  "touch3: $Q --opt mfma_sched=5" \
  "ship_again: $Q" \
  "touch1_again: $Q --opt mfma_sched=3" \
  "touch2_parity: --no-cpu-baseline --no-sub --workload cfg3 --steps 10 --opt mfma_sched=4" 2>&1 | tee $O/variants.txt
