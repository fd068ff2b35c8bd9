#!/bin/bash
# round 2, probe 3: threshold-ladder geometry sweep + PMC counters of variant 6
mkdir -p gpurun_out/r2p3
B="python bench.py --workload cfg3 --no-cpu-baseline --no-parity --steps 5 --warmup 2"
for v in "mfma_sample_rows=131072" "mfma_sample_rows=32768" "mfma_sample_rows=8192" "mfma_sample_rows=4096" "mfma_sample_rows=4096 --opt mfma_ladder=8" "mfma_sample_rows=16384 --opt mfma_ladder=8" \
         "mfma_sample_rows=8192 --opt mfma_v6_min_rows=100000" "mfma_sample_rows=8192 --opt mfma_v6_min_rows=30000" "mfma_sample_rows=8192 --opt mfma_v6_min_rows=0"; do
  echo "== $v" >> gpurun_out/r2p3/cfg3.jsonl
  $B --opt $v >> gpurun_out/r2p3/cfg3.jsonl 2>> gpurun_out/r2p3/cfg3.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p3/cfg3.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.2f  kernel %.2f ms  frac %.4f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac']), r.get('kernel_parts_ms_per_step'), r['kernel_launches_per_step'])
    except Exception as e: print('   ??', l[:200])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/r2p3/pmc_$tag -- python $R/bench.py --workload cfg3 --no-cpu-baseline --no-parity --steps 3 --warmup 1 --opt mfma_variant=6 > $R/gpurun_out/r2p3/pmc_$tag.log 2>&1
done
ls -R $R/gpurun_out/r2p3 | head -30
