#!/bin/bash
# A/B on one box, interleaved: $1 = extra args of variant B (e.g. "--opt mfma_sched=6"), $2 = rounds, $3 = workload args (default cfg3)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/ab; mkdir -p $O
Q="--no-cpu-baseline --no-parity --no-sub ${3:---workload cfg3} --steps 20 --warmup 5"
specs=()
for i in $(seq 1 ${2:-3}); do specs+=("A$i: $Q" "B$i: $Q $1"); done
timeout 1700 python tools/bench_variants.py $O "${specs[@]}" 2>&1 | tee $O/variants.txt
