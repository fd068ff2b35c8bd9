#!/bin/bash
# per-phase kernel durations of the cfg3 batch for MFMA variants 3 and 5 (rocprofv3 kernel trace)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for v in 3 5; do
  rm -rf $OUT/phase_v$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/phase_v$v -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --opt mfma_variant=$v > $OUT/phase_v$v.log 2>&1
  python - $OUT/phase_v$v <<'PY'
import sys,glob,csv
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'mfma_scan' in r['Kernel_Name']]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows]
    print(sys.argv[1].split('/')[-1], [round(x,3) for x in d[-8:]])
PY
done
