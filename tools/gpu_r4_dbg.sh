#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
S="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --rows 1250000 --steps 60 --warmup 10"
for extra in "" "--opt mfma_ladder=4" "" "--opt wide_fallback=1"; do
  echo "== [$extra]"; TAVB_BENCH_DEBUG=1 python bench.py $S $extra 2>&1 >/dev/null | grep -E "bench debug|lookups" ; TAVB_BENCH_DEBUG=1 python bench.py $S $extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['p50_latency_us'], d['p99_latency_us'], d['roofline']['kernel_ms_per_step'])"
done
