#!/bin/bash
# Round-2 evidence for profiles/: bench lines, rocprofv3 kernel traces, the vendor GEMM on the cfg3 shape.  Everything lands in gpurun_out/r02/; the summaries are copied to profiles/r02_* by hand.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
cd $R
echo "== $(date -u +%FT%TZ)" > $O/round.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?" >> $O/round.log
for wl in cfg2_f16 cfg2_b32 cfg2_b1024 cfg3_b32 cfg3_b128 cfg1 cfg4; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?" >> $O/round.log
done
timeout 600 python bench.py --workload cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --workload cfg5 --cfg5-separate > $O/bench_cfg5_separate.json 2> $O/bench_cfg5_separate.err
timeout 300 python tools/gemm_rate.py > $O/gemm_rate.json 2> $O/gemm_rate.err
cd /tmp
Q="--no-cpu-baseline --no-parity"
# kernel traces (the same commands as the bench lines, fewer steps)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_default -o x -- python $R/bench.py $Q --steps 4 --warmup 1 > $O/trace_default.log 2>&1
for wl in cfg2_b32 cfg3_b32 cfg3_b128 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$wl -o x -- python $R/bench.py $Q --workload $wl --steps 4 --warmup 1 > $O/trace_$wl.log 2>&1
done
# (the PMC passes are tools/gpu_r2_pmc.sh: FETCH_SIZE and WRITE_SIZE in ONE pass abort rocprofv3 on gfx950 and hang it for its whole timeout)
cd $R
for t in default cfg2_b32 cfg3_b32 cfg3_b128 cfg5; do
  db=$(find $O/trace_$t -name "*results.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/trace_$t.md 2>> $O/round.log
done
# keep the merge-back small: the summaries, not the raw traces
rm -rf $O/trace_*/
for f in $O/bench_*.json; do echo "--- $(basename $f)"; head -c 1500 $f; echo; done >> $O/round.log
echo "== done $(date -u +%FT%TZ)" >> $O/round.log
tail -n 5 $O/round.log | cut -c1-300
ls $O
