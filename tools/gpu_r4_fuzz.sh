#!/bin/bash
# round 4: the random differential cases with other seeds than the suite's (1000 cases), once
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/fuzz; mkdir -p $O
for base in 2000 3000 4000 5000; do
  TAVB_FUZZ_BASE=$base timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k random_case > $O/fuzz_$base.log 2>&1; echo "base $base rc=$? $(tail -1 $O/fuzz_$base.log)"
done
