#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/inline; mkdir -p $O
export TAVB_BENCH_DEBUG=1
for extra in "" "--opt inline_query=0" "--steps 2000" "--steps 2000 --opt inline_query=0"; do
  echo "== $extra"
  timeout 300 python bench.py --workload cfg1 --no-cpu-baseline --no-parity $extra 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['steps'], d['warmup'], d['ms_per_step'], d.get('p50_latency_us'), d.get('p99_latency_us'), d.get('min_latency_us'))"
  grep "bench debug" $O/err.txt
done
