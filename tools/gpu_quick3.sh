#!/bin/bash
# quick cfg3 check: parity subset + bench line(s); extra bench options via "$@" (one run per quoted argument)
mkdir -p gpurun_out
[ -z "$SKIP_TESTS" ] && timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "mfma" -p no:cacheprovider -x 2>&1 | tail -3
runs=("$@"); [ ${#runs[@]} -eq 0 ] && runs=("")
for opts in "${runs[@]}"; do
  echo "== $opts"
  timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 5 --warmup 2 $opts 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), d['ms_per_step'], r['achieved'], r['frac'], r['kernel_avg_ms'])"
done 2>&1 | tee gpurun_out/quick3.log
