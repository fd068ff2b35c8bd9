#!/bin/bash
# round 4, opening run: headline alone + the ablations of the wide tile on the same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/base; mkdir -p $O
Q="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 20 --warmup 5"
timeout 1200 python tools/bench_variants.py $O \
  "ship: $Q" \
  "abl258_mfma_only: $Q --opt mfma_ablate=258" \
  "abl256_no_admit: $Q --opt mfma_ablate=256" \
  "abl268_cache_res: $Q --opt mfma_ablate=268" \
  "ship_again: $Q" 2>&1 | tee $O/variants.txt
rocm-smi --showpower --showclocks 2>/dev/null | head -30 > $O/smi.txt
