#!/bin/bash
# round 4: what do admissions cost on a 1/8 shard?  (ablation 256 = no admission test at all: the floor of any threshold improvement)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/adm; mkdir -p $O
Q="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --rows 1250000 --steps 40 --warmup 10"
timeout 1500 python tools/bench_variants.py $O \
  "shard: $Q" \
  "shard_abl256: $Q --opt mfma_ablate=256" \
  "shard_abl258: $Q --opt mfma_ablate=258" \
  "shard_again: $Q" \
  "shard_abl256_again: $Q --opt mfma_ablate=256" \
  "shard_onephase_abl256: $Q --opt mfma_ablate=256 --opt mfma_sample_rows=-1" \
  "shard_ms07: $Q --min-score 0.57" 2>&1 | tee $O/variants.txt
