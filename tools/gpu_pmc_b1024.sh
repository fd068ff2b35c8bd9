export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $OUT/prof_final_cfg2_b1024 $OUT/pmc_b1024
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_final_cfg2_b1024 -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2_b1024 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_final_cfg2_b1024.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_b1024 -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2_b1024 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_b1024.log 2>&1
python - $OUT/pmc_b1024 <<'PY'
import sys,glob,csv,collections
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'skinny' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k,len(v),sum(v)/len(v))
PY
