#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
for A in 34; do
  timeout 900 python bench.py --workload cfg3 --rows 4000000 --steps 4 --warmup 1 --no-cpu-baseline --opt mfma_variant=4 --opt mfma_ablate=$A > $OUT/ab4_$A.json 2> $OUT/ab4_$A.err
  python -c "
import json;d=json.load(open('$OUT/ab4_$A.json'));print('v4 ablate $A kernel_ms', round(d['roofline']['kernel_avg_ms'],3), 'TF-eq', round(d['roofline']['achieved'],1))" | tee -a $OUT/round.log
done
cd /tmp
for A in 34 2 0; do
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_v4_$A -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --rows 4000000 --steps 2 --warmup 1 --no-cpu-baseline --opt mfma_variant=4 --opt mfma_ablate=$A > $OUT/pmc_v4_$A.log 2>&1
python - <<PY | tee -a $OUT/round.log
import csv,collections
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for row in csv.DictReader(open('$OUT/pmc_v4_$A/x_counter_collection.csv')):
    if 'mfma_scan' in row['Kernel_Name']:
        a=agg[row['Counter_Name']]; a[0]+=1; a[1]+=float(row['Counter_Value']); a[2]+=(int(row['End_Timestamp'])-int(row['Start_Timestamp']))
for k,(n,v,t) in agg.items(): print('ablate $A', k, 'per-dispatch', v/n, 'dur_ms', t/n/1e6, 'GHz(if GRBM/8)', v/n/8/(t/n) if k=='GRBM_GUI_ACTIVE' else '')
PY
done
echo "== done" | tee -a $OUT/round.log
