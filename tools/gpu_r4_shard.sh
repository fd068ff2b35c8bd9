#!/bin/bash
# round 4: a 1/8 shard of cfg3 on one GPU, with parity (the projected per-GPU step of the 8-GPU strong-scaling run), default path and mfma_bdirect=1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/shard; mkdir -p $O
S="--no-cpu-baseline --no-sub --no-calibration --workload cfg3 --rows 1250000 --steps 60 --warmup 10"
F="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --steps 20 --warmup 5"
timeout 1700 python tools/bench_variants.py $O "shard_1of8: $S" "shard_1of8_bdirect: $S --opt mfma_bdirect=1" "full: $F" "full_bdirect: $F --opt mfma_bdirect=1" "shard_1of8_again: $S --no-parity" "shard_1of8_bdirect_again: $S --no-parity --opt mfma_bdirect=1" 2>&1 | tee $O/variants.txt
