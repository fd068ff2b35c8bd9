#!/bin/bash
# round 4: where does the LDS-DMA's cost go -- clock or stalls?  PMC pass (clock = GRBM_GUI_ACTIVE / 8 / duration, pipe busy = MFMA_BUSY / (1024 SIMDs x clocks)) per ablation mode
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4/pmc_abl; rm -rf $O; mkdir -p $O
cd /tmp
Q="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 2 --warmup 1"
for m in 0 256 268 258; do
  timeout -k 5 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/raw_$m -- python $R/bench.py $Q --opt mfma_ablate=$m > $O/pmc_$m.log 2>&1
  echo "abl $m rc=$?"
  python $R/tools/pmc_summary.py $O/raw_$m/*/*_counter_collection.csv --steps 1 --cmd "bench.py $Q --opt mfma_ablate=$m" 2>/dev/null | grep -E "^\| kernel|mfma_scan" > $O/pmc_$m.md
  rm -rf $O/raw_$m
  cat $O/pmc_$m.md
done
