#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/share2; mkdir -p $O
S="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --rows 1250000 --steps 40 --warmup 10"
timeout 1500 python tools/bench_variants.py $O \
  "shard: $S" "union8_same_ladder: $S --opt share_emulate_world=108" "union8_short_ladder: $S --opt share_emulate_world=8" \
  "shard_again: $S" "union8_same_ladder_again: $S --opt share_emulate_world=108" \
  "union8_parity: --no-cpu-baseline --no-sub --workload cfg3 --rows 1250000 --steps 20 --opt share_emulate_world=108" 2>&1 | tee $O/variants.txt
