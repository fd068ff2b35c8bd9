#!/bin/bash
# round 4: the whole GPU suite, smoke, the driver's bench invocation
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/full; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_r4_bench.sh 2>&1 | sed 's#gpurun_out/r4/bench#gpurun_out/r4/bench#'
