#!/bin/bash
# L2 behaviour of the cfg3 MFMA kernel with and without the tile rendezvous (PMC passes, one counter group each)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
for tag in base rdv; do
  opts=""; [ $tag = rdv ] && opts="--opt mfma_rendezvous=1"
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    name=$(echo $grp | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/l2_${tag}_$name -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline $opts > $OUT/l2_${tag}_$name.log 2>&1
    python - "$OUT/l2_${tag}_$name" "$tag" <<'PY'
import sys,glob,csv,collections
d,tag=sys.argv[1],sys.argv[2]
for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mfma_scan' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(tag,k,len(v),sum(v)/len(v))
PY
  done
done 2>&1 | tee $OUT/l2.log
