#!/bin/bash
mkdir -p gpurun_out/r2p17; rm -f gpurun_out/r2p17/*
B="python bench.py --no-cpu-baseline --no-parity --workload cfg2_b32"
for lib in libtavb.so libtavb_ring3.so libtavb_ring4.so; do
 for v in "" "--rows 4000000" "--opt mfma_splits=240" ; do
  echo "== $lib $v" >> gpurun_out/r2p17/b.jsonl
  TAVB_LIBRARY=$lib $B $v >> gpurun_out/r2p17/b.jsonl 2>> gpurun_out/r2p17/b.err
 done
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p17/b.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.3f  kernel %.3f ms  %s frac %.4f  achieved %.1f %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['bound'], r['frac'], r['achieved'], r['unit']), {k: round(v,3) for k,v in r.get('kernel_parts_ms_per_step').items()}, r['kernel_launches_per_step'])
    except Exception as e: print('   ??', l[:200])
PY
tail -n 3 gpurun_out/r2p17/b.err
