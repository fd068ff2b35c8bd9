set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6z; mkdir -p $O
Q="--no-cpu-baseline --no-parity --no-sub --no-calibration"
cd /tmp
for spec in "c2b32:--workload cfg2_b32 --steps 20 --warmup 5" "shard:--workload cfg3 --rows 1250000 --steps 20 --warmup 5" ; do
  n=${spec%%:*}; a=${spec#*:}
  timeout 600 rocprofv3 --kernel-trace -d $O/tr_$n -o x -- python $R/bench.py $Q $a > $O/tl_$n.log 2>&1; echo "$n rc=$?"
  db=$(find $O/tr_$n -name "*results.db" | head -1)
  python $R/tools/rocpd_timeline.py $db --skip 8 > $O/timeline_$n.md; rm -rf $O/tr_$n
  grep -o '"ms_per_step": [0-9.]*' $O/tl_$n.log | head -1
done
