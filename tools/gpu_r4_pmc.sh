#!/bin/bash
# Round-4 evidence for profiles/: PMC passes (one counter group per run), kernel traces of the driver's command, power/clock trace.
# Summaries -> gpurun_out/r4/prof/*.md, traffic per step -> gpurun_out/r4/prof/pmc_traffic.json (copied to profiles/ by hand)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4/prof
rm -rf $O; mkdir -p $O
cd /tmp
Q="--no-cpu-baseline --no-parity --no-sub --no-calibration"
pmc() { # name workload-args counters...
  local name=$1 wl="$2"; shift 2
  timeout -k 5 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/raw_$name -- python $R/bench.py $Q $wl > $O/pmc_$name.log 2>&1
  echo "$name rc=$? $(date -u +%T)" >> $O/round.log
  local total=$(grep -o "[0-9]* lookups in this process" $O/pmc_$name.log | head -1 | cut -d" " -f1)  # bench.py says how many lookups the run made
  python $R/tools/pmc_summary.py $O/raw_$name/*/*_counter_collection.csv --steps ${total:-1} --cmd "bench.py $Q $wl" ${PMC_NAME:+--json $O/pmc_traffic.json --name $PMC_NAME} > $O/pmc_$name.md 2>> $O/round.log
  rm -rf $O/raw_$name
}
PMC_NAME=cfg3 pmc cfg3_fetch "--workload cfg3 --steps 2 --warmup 1" FETCH_SIZE
PMC_NAME= pmc cfg3_mfma "--workload cfg3 --steps 2 --warmup 1" GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
PMC_NAME=cfg3_clustered pmc cfg3_clustered_fetch "--workload cfg3_clustered --steps 2 --warmup 1" FETCH_SIZE
PMC_NAME=cfg3_q1 pmc cfg3_q1_fetch "--workload cfg3_q1 --steps 5 --warmup 1" FETCH_SIZE
PMC_NAME=cfg2 pmc cfg2_fetch "--workload cfg2 --steps 10 --warmup 2" FETCH_SIZE
PMC_NAME=cfg4 pmc cfg4_fetch "--workload cfg4 --steps 2 --warmup 1" FETCH_SIZE
PMC_NAME=cfg3_b128 pmc cfg3_b128_fetch "--workload cfg3_b128 --steps 2 --warmup 1" FETCH_SIZE
PMC_NAME=cfg2_b32 pmc cfg2_b32_fetch "--workload cfg2_b32 --steps 5 --warmup 1" FETCH_SIZE
PMC_NAME=cfg3_b32 pmc cfg3_b32_fetch "--workload cfg3_b32 --steps 2 --warmup 1" FETCH_SIZE
# kernel traces: the driver's command with fewer steps; the headline alone with the driver's steps
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_default -o x -- python $R/bench.py --no-cpu-baseline --no-parity --steps 4 --warmup 1 > $O/trace_default.log 2>&1; echo "trace default rc=$?" >> $O/round.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_headline -o x -- python $R/bench.py $Q --steps 20 --warmup 5 > $O/trace_headline.log 2>&1; echo "trace headline rc=$?" >> $O/round.log
cd $R
for t in default headline; do
  db=$(find $O/trace_$t -name "*results.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/trace_$t.md 2>> $O/round.log
done
rm -rf $O/trace_*/
grep -o '"kernel_ms_per_step":[0-9.]*' $O/trace_headline.log | head -2
# power / clock under the cfg3 loop
python bench.py --workload cfg3 $Q --steps 400 --warmup 2 > $O/power_bench.json 2> $O/power_bench.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "GPU\[0\].*(Power|sclk)" | tr '\n' ' ' >> $O/smi.txt; echo >> $O/smi.txt; sleep 0.4
done
rocm-smi --showmaxpower 2>/dev/null | grep -E "GPU\[0\]" >> $O/smi.txt
cat $O/round.log; tail -3 $O/smi.txt
python -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print({k:(round(v['traffic_bytes_per_step']/1e9,3), v['launches_per_step']) for k,v in d.items()})"
