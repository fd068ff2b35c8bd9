#!/bin/bash
# Template-ablation timing of the MFMA kernel (garbage results in ablation modes).  Usage: edit the list below.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
for CFG in "3 0 131072" "3 256 131072" "3 0 0" "3 256 0"; do
  set -- $CFG
  timeout 900 python bench.py --workload cfg3 --steps 4 --warmup 1 --no-cpu-baseline --opt mfma_variant=$1 --opt mfma_ablate=$2 --opt mfma_sample_rows=$3 > $OUT/abx.json 2> $OUT/abx.err
  python -c "
import json;d=json.load(open('$OUT/abx.json'));print('variant $1 ablate $2 sample $3 kernel_ms', round(d['roofline']['kernel_avg_ms'],3), 'TF-eq', round(d['roofline']['achieved'],1))" | tee -a $OUT/round.log
done
echo "== done" | tee -a $OUT/round.log
