#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
echo "== pytest gpu (mfma + fixed tests)" | tee -a $OUT/round.log
timeout 900 python -m pytest tests -m gpu -q -n 1 --tb=short -p no:cacheprovider -k "mfma or threshold_rule or golden_seeded" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round.log
tail -15 $OUT/pytest_gpu.log | tee -a $OUT/round.log

echo "== bench cfg3 (2M rows first)" | tee -a $OUT/round.log
timeout 600 python bench.py --workload cfg3 --rows 2000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg3_2m.json 2> $OUT/bench_cfg3_2m.err
echo "rc=$?" | tee -a $OUT/round.log
cat $OUT/bench_cfg3_2m.json | tee -a $OUT/round.log
tail -3 $OUT/bench_cfg3_2m.err | tee -a $OUT/round.log

echo "== bench cfg3 full" | tee -a $OUT/round.log
timeout 900 python bench.py --workload cfg3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
echo "rc=$?" | tee -a $OUT/round.log
cat $OUT/bench_cfg3.json | tee -a $OUT/round.log
tail -3 $OUT/bench_cfg3.err | tee -a $OUT/round.log

echo "== sweep f16 single query" | tee -a $OUT/round.log
timeout 600 python tools/sweep_scan.py --tag f16 --dtype fp16 --rows 2000000 --quick > $OUT/sweep_f16.log 2>&1
echo "sweep rc=$?" | tee -a $OUT/round.log
tail -1 $OUT/sweep_f16.log | tee -a $OUT/round.log
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
