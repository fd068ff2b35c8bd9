#!/bin/bash
# Round-2 evidence for profiles/: bench lines, rocprofv3 kernel traces, PMC passes (FETCH/WRITE, matrix-pipe busy, LDS), the
# vendor GEMM on the cfg3 shape.  Everything lands in gpurun_out/r02/; the summaries are copied to profiles/r02_* by hand.
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
# PMC passes: one counter group per run
pmc() { # name steps+warmup workload-args counters...
  local name=$1 total=$2 wl="$3"; shift 3
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- python $R/bench.py $Q $wl > $O/pmc_$name.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$name/*/*_counter_collection.csv --steps $total ${PMC_JSON:+--json $PMC_JSON --name $PMC_NAME} > $O/pmc_$name.md 2>> $O/round.log
}
PMC_JSON=$O/pmc_traffic.json
PMC_NAME=cfg3 pmc cfg3_fetch 3 "--workload cfg3 --steps 2 --warmup 1" FETCH_SIZE WRITE_SIZE
PMC_NAME=cfg3_q1 pmc cfg3_q1_fetch 6 "--workload cfg3_q1 --steps 5 --warmup 1" FETCH_SIZE WRITE_SIZE
PMC_NAME=cfg2 pmc cfg2_fetch 12 "--workload cfg2 --steps 10 --warmup 2" FETCH_SIZE WRITE_SIZE
PMC_NAME=cfg2_b32 pmc cfg2_b32_fetch 6 "--workload cfg2_b32 --steps 5 --warmup 1" FETCH_SIZE WRITE_SIZE
PMC_NAME=cfg3_b32 pmc cfg3_b32_fetch 3 "--workload cfg3_b32 --steps 2 --warmup 1" FETCH_SIZE WRITE_SIZE
PMC_NAME=cfg3_b128 pmc cfg3_b128_fetch 3 "--workload cfg3_b128 --steps 2 --warmup 1" FETCH_SIZE WRITE_SIZE
unset PMC_JSON
pmc cfg3_mfma 3 "--workload cfg3 --steps 2 --warmup 1" GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
pmc cfg3_lds 3 "--workload cfg3 --steps 2 --warmup 1" SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM
pmc cfg3_l2 3 "--workload cfg3 --steps 2 --warmup 1" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pmc cfg3_b128_mfma 3 "--workload cfg3_b128 --steps 2 --warmup 1" GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
cd $R
for t in default cfg2_b32 cfg3_b32 cfg3_b128 cfg5; do
  db=$(find $O/trace_$t -name "*results.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/trace_$t.md 2>> $O/round.log
done
# keep the merge-back small: the summaries, not the raw traces
rm -rf $O/trace_*/ $O/pmc_*/
for f in $O/bench_*.json; do echo "--- $(basename $f)"; head -c 1500 $f; echo; done >> $O/round.log
echo "== done $(date -u +%FT%TZ)" >> $O/round.log
tail -n 5 $O/round.log | cut -c1-300
ls $O
