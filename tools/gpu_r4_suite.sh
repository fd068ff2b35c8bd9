#!/bin/bash
# round 4: the whole GPU suite + the mid-batch / skinny-tile workloads after the admission path change
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/suite; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.log
timeout 1500 python tools/bench_variants.py $O \
  "cfg2_b32: --workload cfg2_b32 --no-cpu-baseline" "cfg3_b32: --workload cfg3_b32 --no-cpu-baseline" "cfg3_b128: --workload cfg3_b128 --no-cpu-baseline" \
  "cfg2_b1024: --workload cfg2_b1024 --no-cpu-baseline" "cfg5: --workload cfg5" "cfg1: --workload cfg1 --no-cpu-baseline" 2>&1 | tee $O/variants.txt
