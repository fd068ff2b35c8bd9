#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
timeout 1500 python -m pytest tests -m gpu -q -n 1 --tb=short -p no:cacheprovider -k "fused or cfg3_full_size" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round.log
tail -25 $OUT/pytest_gpu.log | tee -a $OUT/round.log
for MODE in "" "--cfg5-separate" "--cfg5-subset 1000" "--cfg5-subset 1000 --cfg5-separate"; do
  TAG=$(echo "fused$MODE" | tr -d ' -')
  timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 3 $MODE > $OUT/bench_cfg5_$TAG.json 2> $OUT/bench_cfg5_$TAG.err
  echo "cfg5 $MODE rc=$?" | tee -a $OUT/round.log
  python -c "
import json;d=json.load(open('$OUT/bench_cfg5_$TAG.json'));print(d['value'], d['ms_per_step'], d['p50_latency_us'], d['roofline'])" | tee -a $OUT/round.log
  tail -2 $OUT/bench_cfg5_$TAG.err | tee -a $OUT/round.log
done
timeout 600 python bench.py --workload cfg1 --steps 200 --warmup 20 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err
cat $OUT/bench_cfg1.json | tee -a $OUT/round.log
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
