#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/small3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_corpus or golden_explicit or known_answer or float32_threshold or tiers or cursor" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for spec in "10000 10 0.0" "1294 50 0.85" "1294 10 0.0"; do
  timeout 300 python tools/latency_breakdown.py $spec 2>&1 | grep -E "^rows|ONE launch|two launches|device-resident|VectorBase" 
done
timeout 600 python tools/bench_variants.py $O "cfg1: --workload cfg1 --no-cpu-baseline --class-api" 2>&1 | tee $O/variants.txt
