#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
for A in 33 97 225 0 64 192; do
  timeout 900 python bench.py --workload cfg3 --rows 4000000 --steps 4 --warmup 1 --no-cpu-baseline --opt mfma_variant=4 --opt mfma_ablate=$A > $OUT/ab4_$A.json 2> $OUT/ab4_$A.err
  python -c "
import json;d=json.load(open('$OUT/ab4_$A.json'));print('v4 ablate $A kernel_ms', round(d['roofline']['kernel_avg_ms'],3), 'TF-eq', round(d['roofline']['achieved'],1))" | tee -a $OUT/round.log
done
echo "== done" | tee -a $OUT/round.log
