#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
timeout 900 python -m pytest tests -m gpu -q -n 1 --tb=short -p no:cacheprovider -k "mfma" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round.log
tail -15 $OUT/pytest_gpu.log | tee -a $OUT/round.log
for V in 4 3; do
  timeout 900 python bench.py --workload cfg3 --steps 5 --warmup 1 --no-cpu-baseline --opt mfma_variant=$V > $OUT/bench_cfg3_v$V.json 2> $OUT/bench_cfg3_v$V.err
  python -c "
import json;d=json.load(open('$OUT/bench_cfg3_v$V.json'));print('variant $V', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel_avg_ms'])" | tee -a $OUT/round.log
  tail -2 $OUT/bench_cfg3_v$V.err | tee -a $OUT/round.log
done
for A in 1 2 3 4; do
  timeout 900 python bench.py --workload cfg3 --rows 4000000 --steps 4 --warmup 1 --no-cpu-baseline --opt mfma_variant=4 --opt mfma_ablate=$A > $OUT/ab4_$A.json 2> $OUT/ab4_$A.err
  python -c "
import json;d=json.load(open('$OUT/ab4_$A.json'));print('v4 ablate $A kernel_ms', round(d['roofline']['kernel_avg_ms'],3), 'TF-eq', round(d['roofline']['achieved'],1))" | tee -a $OUT/round.log
done
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
