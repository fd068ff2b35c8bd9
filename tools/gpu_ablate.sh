#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
for CFG in "2 0" "2 1" "2 2" "0 0" "0 1" "0 2"; do
  set -- $CFG
  timeout 900 python bench.py --workload cfg3 --rows 4000000 --steps 4 --warmup 1 --no-cpu-baseline --opt mfma_variant=2 --opt mfma_ablate=$1 --opt mfma_group=$2 > $OUT/ab_$1_$2.json 2> $OUT/ab_$1_$2.err
  python -c "
import json;d=json.load(open('$OUT/ab_$1_$2.json'));print('ablate $1 group $2 kernel_ms', round(d['roofline']['kernel_avg_ms'],3), 'TF-eq', round(d['roofline']['achieved'],1))" | tee -a $OUT/round.log
done
echo "== done" | tee -a $OUT/round.log
