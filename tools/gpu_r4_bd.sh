#!/bin/bash
# round 4: the 256-query tile with its query operand straight from L2 (fragment-major layout, no LDS staging): correctness, then A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/bd; mkdir -p $O
F="--no-cpu-baseline --no-sub --no-calibration --workload cfg3 --steps 20 --warmup 5"
S="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --rows 1250000 --steps 60 --warmup 10"
timeout 1700 python tools/bench_variants.py $O "bd_parity: $F --opt mfma_bdirect=1" "base1: $F --no-parity" "bd1: $F --no-parity --opt mfma_bdirect=1" "base2: $F --no-parity" "bd2: $F --no-parity --opt mfma_bdirect=1" \
  "s_base: $S" "s_bd: $S --opt mfma_bdirect=1" 2>&1 | tee $O/variants.txt
