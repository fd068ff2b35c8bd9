#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
timeout 1500 python -m pytest tests -m gpu -q -n 2 --tb=short -p no:cacheprovider -k "not cfg3_full and not one_million" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round.log
tail -8 $OUT/pytest_gpu.log | tee -a $OUT/round.log
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 3 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python -c "
import json;d=json.load(open('$OUT/bench_cfg5.json'));print('cfg5', d['value'], d['p50_latency_us'], d['roofline']['scan_kernel_ms_per_user_query'])" | tee -a $OUT/round.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python -c "
import json;d=json.load(open('$OUT/bench_cfg2.json'));print('cfg2', d['value'], d['p50_latency_us'], d['roofline']['achieved'], d['roofline']['merge_avg_us'])" | tee -a $OUT/round.log
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
