#!/bin/bash
# round 4: the query inside the kernel arguments of the one-launch small-corpus lookup -- tests, latency breakdown, cfg1 bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/inline; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_corpus or golden or differential or tiers or geometry or captured_graph or profile_counters" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for spec in "10000 10 0.0" "1294 50 0.85" "1294 10 0.0"; do
  timeout 300 python tools/latency_breakdown.py $spec > $O/latency_$(echo $spec | tr ' ' '_').txt 2>&1; grep -E "^rows|ONE launch|kernel arguments|inline_query|two launches|fuzzy_lookup|kernel scan" $O/latency_$(echo $spec | tr ' ' '_').txt
done
timeout 600 python tools/bench_variants.py $O "cfg1: --workload cfg1 --no-cpu-baseline --class-api" "cfg1_copy: --workload cfg1 --no-cpu-baseline --class-api --opt inline_query=0" 2>&1 | tee $O/variants.txt
