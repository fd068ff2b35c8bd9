#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
timeout 1500 python -m pytest tests -m gpu -q -n 2 --tb=short -p no:cacheprovider -k "mfma or k_blocked" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round.log
tail -12 $OUT/pytest_gpu.log | tee -a $OUT/round.log
for V in 5 3 4; do
timeout 900 python bench.py --workload cfg3 --steps 5 --warmup 1 --no-cpu-baseline --opt mfma_variant=$V > $OUT/bench_cfg3_v$V.json 2> $OUT/bench_cfg3_v$V.err
python -c "
import json;d=json.load(open('$OUT/bench_cfg3_v$V.json'));print('cfg3 variant $V', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel_avg_ms'])" | tee -a $OUT/round.log
tail -1 $OUT/bench_cfg3_v$V.err | tee -a $OUT/round.log
done
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
