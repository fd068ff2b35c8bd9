#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/dup3; mkdir -p $O
S="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --rows 1250000 --steps 60 --warmup 10"
TAVB_BENCH_DEBUG=1 timeout 1700 python tools/bench_variants.py $O "s_nofb1: $S --opt wide_fallback=0" "s_fb1: $S" "s_nofb2: $S --opt wide_fallback=0" "s_fb2: $S --opt wide_fallback=1" "s_other_opt: $S --opt mfma_ladder=4" 2>&1 | tee $O/variants.txt
