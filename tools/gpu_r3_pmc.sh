#!/bin/bash
# Round-3 evidence for profiles/: PMC passes (one counter group per run; FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950), kernel traces,
# the vendor GEMM on the cfg3 shape, power/clock trace.  Summaries -> gpurun_out/r3/prof/*.md, traffic per step -> gpurun_out/r3/prof/pmc_traffic.json
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/prof
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "wide_tile or clustered or skinny or captured_graph" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
cd /tmp
Q="--no-cpu-baseline --no-parity"
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
# kernel traces (the driver's command with fewer steps; cfg2_b32; cfg5)
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_default -o x -- python $R/bench.py $Q --steps 4 --warmup 1 > $O/trace_default.log 2>&1; echo "trace default rc=$?" >> $O/round.log
for wl in cfg2_b32 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$wl -o x -- python $R/bench.py $Q --workload $wl --steps 4 --warmup 1 > $O/trace_$wl.log 2>&1
done
cd $R
for t in default cfg2_b32 cfg5; do
  db=$(find $O/trace_$t -name "*results.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/trace_$t.md 2>> $O/round.log
done
rm -rf $O/trace_*/
# the vendor GEMM on the cfg3 shape and the power/clock trace of the cfg3 loop, same box
timeout 300 python tools/gemm_rate.py > $O/gemm_rate.json 2> $O/gemm_rate.err
python bench.py --workload cfg3 $Q --steps 400 --warmup 2 > $O/power_bench.json 2> $O/power_bench.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "GPU\[0\].*(Power|sclk)" | tr '\n' ' ' >> $O/smi.txt; echo >> $O/smi.txt; sleep 0.4
done
rocm-smi --showmaxpower 2>/dev/null | grep -E "GPU\[0\]" >> $O/smi.txt
# bench lines of the other workloads
timeout 1500 python tools/bench_variants.py $O \
  "cfg2_b32: --workload cfg2_b32 --no-cpu-baseline" "cfg2_f16: --workload cfg2_f16 --no-cpu-baseline" "cfg1: --workload cfg1 --no-cpu-baseline" \
  "cfg3_b32: --workload cfg3_b32 --no-cpu-baseline" "cfg3_b128: --workload cfg3_b128 --no-cpu-baseline" "cfg2_b1024: --workload cfg2_b1024 --no-cpu-baseline" \
  "cfg5_separate: --workload cfg5 --cfg5-separate" "cfg5_subset1000: --workload cfg5 --cfg5-subset 1000" 2>&1 | tee $O/variants.txt
cat $O/round.log; cat $O/gemm_rate.json; tail -3 $O/smi.txt; python -c "
import json; d=json.loads(open('$O/power_bench.json').read()); print('power run', d['ms_per_step'], d['roofline']['frac'])"
python -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print({k:(round(v['traffic_bytes_per_step']/1e9,3), v['kernels']) for k,v in d.items()})"
