#!/bin/bash
mkdir -p gpurun_out/r2p16
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "skinny or batch or wide or fused or texts" > gpurun_out/r2p16/pytest.txt 2>&1
tail -n 6 gpurun_out/r2p16/pytest.txt
B="python bench.py --no-cpu-baseline --no-parity"
for v in "--workload cfg2_b32" "--workload cfg2_b32 --opt mfma_sched=9" "--workload cfg2_b1024" "--workload cfg2_b1024 --opt mfma_sched=9" "--workload cfg3_b32" "--workload cfg3_b32 --opt mfma_sched=9" \
         "--workload cfg3_b32 --opt skinny_min_batch_f16=100" "--workload cfg3_b128" "--workload cfg3_b128 --opt mfma_min_batch=1000" "--workload cfg3_b128 --opt mfma_min_batch=1000 --opt mfma_sched=9"; do
  echo "== $v" >> gpurun_out/r2p16/b.jsonl
  $B $v >> gpurun_out/r2p16/b.jsonl 2>> gpurun_out/r2p16/b.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p16/b.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.3f  kernel %.3f ms  %s frac %.4f  achieved %.1f %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['bound'], r['frac'], r['achieved'], r['unit']), {k: round(v,3) for k,v in r.get('kernel_parts_ms_per_step').items()}, r['kernel_launches_per_step'])
    except Exception as e: print('   ??', l[:200])
PY
tail -n 3 gpurun_out/r2p16/b.err
