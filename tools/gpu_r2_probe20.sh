#!/bin/bash
mkdir -p gpurun_out/r2p20; rm -f gpurun_out/r2p20/*
B="python bench.py --no-cpu-baseline --no-parity --workload cfg3 --steps 20 --warmup 3"
for v in "--rows 1250000" "--rows 1250000 --opt mfma_ladder=6" "--rows 1250000 --opt mfma_sample_rows=-1" "--rows 2500000" "--rows 5000000"; do
  echo "== $v" >> gpurun_out/r2p20/b.jsonl
  timeout 600 $B $v >> gpurun_out/r2p20/b.jsonl 2>> gpurun_out/r2p20/b.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p20/b.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.3f  kernel %.3f ms  %s frac %.4f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['bound'], r['frac']), {k: round(v,3) for k,v in r.get('kernel_parts_ms_per_step').items()}, 'other', {k: round(v,3) for k,v in r.get('other_kernels_ms_per_step',{}).items()}, r['kernel_launches_per_step'])
    except Exception as e: print('   ??', l[:200])
PY
