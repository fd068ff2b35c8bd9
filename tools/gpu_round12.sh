#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
timeout 1500 python -m pytest tests -m gpu -q -n 2 --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round.log
tail -12 $OUT/pytest_gpu.log | tee -a $OUT/round.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python -c "
import json;d=json.load(open('$OUT/bench_cfg2.json'));print('cfg2', d['value'], d['p50_latency_us'], d['roofline']['achieved'], d['roofline']['merge_avg_us'])" | tee -a $OUT/round.log
timeout 600 python bench.py --workload cfg1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err
python -c "
import json;d=json.load(open('$OUT/bench_cfg1.json'));print('cfg1', d['value'], d['p50_latency_us'], d['roofline']['kernel_avg_ms'], d['roofline']['merge_avg_us'])" | tee -a $OUT/round.log
timeout 900 python bench.py --workload cfg3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python -c "
import json;d=json.load(open('$OUT/bench_cfg3.json'));print('cfg3', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['merge_avg_us'])" | tee -a $OUT/round.log
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
