#!/bin/bash
# Round-2 PMC passes (one counter group per run; FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: "exceeds the
# capabilities of the hardware").  Summaries -> gpurun_out/r02pmc/*.md, traffic per step -> gpurun_out/r02pmc/pmc_traffic.json
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02pmc
rm -rf $O; mkdir -p $O
cd /tmp
Q="--no-cpu-baseline --no-parity"
pmc() { # name (unused) workload-args counters...
  local name=$1 total=$2 wl="$3"; shift 3
  timeout -k 5 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/raw_$name -- python $R/bench.py $Q $wl > $O/pmc_$name.log 2>&1
  echo "$name rc=$? $(date -u +%T)" >> $O/round.log
  total=$(grep -o "[0-9]* lookups in this process" $O/pmc_$name.log | head -1 | cut -d" " -f1)  # bench.py says how many lookups the run made
  python $R/tools/pmc_summary.py $O/raw_$name/*/*_counter_collection.csv --steps ${total:-1} --cmd "bench.py $Q $wl" ${PMC_NAME:+--json $O/pmc_traffic.json --name $PMC_NAME} > $O/pmc_$name.md 2>> $O/round.log
  rm -rf $O/raw_$name
}
PMC_NAME=cfg3 pmc cfg3_fetch 3 "--workload cfg3 --steps 2 --warmup 1" FETCH_SIZE
PMC_NAME= pmc cfg3_mfma 3 "--workload cfg3 --steps 2 --warmup 1" GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
PMC_NAME=cfg3_q1 pmc cfg3_q1_fetch 6 "--workload cfg3_q1 --steps 5 --warmup 1" FETCH_SIZE
PMC_NAME=cfg2 pmc cfg2_fetch 12 "--workload cfg2 --steps 10 --warmup 2" FETCH_SIZE
PMC_NAME=cfg3_b128 pmc cfg3_b128_fetch 3 "--workload cfg3_b128 --steps 2 --warmup 1" FETCH_SIZE
PMC_NAME=cfg2_b32 pmc cfg2_b32_fetch 6 "--workload cfg2_b32 --steps 5 --warmup 1" FETCH_SIZE
PMC_NAME=cfg3_b32 pmc cfg3_b32_fetch 3 "--workload cfg3_b32 --steps 2 --warmup 1" FETCH_SIZE
PMC_NAME= pmc cfg3_lds 3 "--workload cfg3 --steps 2 --warmup 1" SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM
PMC_NAME= pmc cfg3_b128_mfma 3 "--workload cfg3_b128 --steps 2 --warmup 1" GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
PMC_NAME= pmc cfg3_write 3 "--workload cfg3 --steps 2 --warmup 1" WRITE_SIZE
cat $O/round.log; cat $O/pmc_traffic.json 2>/dev/null | head -40
