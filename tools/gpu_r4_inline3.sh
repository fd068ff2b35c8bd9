#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/inline; mkdir -p $O
export TAVB_BENCH_DEBUG=1
timeout 900 python tools/bench_variants.py $O "cfg1_copy: --workload cfg1 --no-cpu-baseline --class-api --opt inline_query=0" "cfg1: --workload cfg1 --no-cpu-baseline --class-api" "cfg1_again: --workload cfg1 --no-cpu-baseline --class-api" "cfg1_noclass: --workload cfg1 --no-cpu-baseline" 2>&1 | tee $O/variants.txt
