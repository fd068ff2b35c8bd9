#!/bin/bash
# What the driver runs at round end, timed.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
S=$(date +%s); timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_driver.log 2>&1; echo "pytest rc=$? secs=$(( $(date +%s)-S ))" | tee -a $OUT/round.log
tail -5 $OUT/pytest_driver.log | tee -a $OUT/round.log
S=$(date +%s); timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? secs=$(( $(date +%s)-S ))" | tee -a $OUT/round.log
tail -2 $OUT/smoke.log | tee -a $OUT/round.log
S=$(date +%s); timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? secs=$(( $(date +%s)-S ))" | tee -a $OUT/round.log
cat $OUT/bench_default.json | cut -c1-1200 | tee -a $OUT/round.log
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 50 --warmup 5 > $OUT/bench_k50.json 2> $OUT/bench_k50.err; echo "bench2 rc=$? secs=$(( $(date +%s)-S ))" | tee -a $OUT/round.log
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
