#!/bin/bash
mkdir -p gpurun_out/r2p19; rm -f gpurun_out/r2p19/*
B="python bench.py --no-cpu-baseline --no-parity --workload cfg3 --steps 6"
for nq in 192 384 640; do for t in 128 256; do
  echo "== nq $nq tile $t" >> gpurun_out/r2p19/b.jsonl
  timeout 600 $B --queries $nq --opt mfma_tile=$t >> gpurun_out/r2p19/b.jsonl 2>> gpurun_out/r2p19/b.err
done; done
python - <<'PY'
import json
for l in open('gpurun_out/r2p19/b.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.3f  kernel %.3f ms  %s frac %.4f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['bound'], r['frac']), {k: round(v,3) for k,v in r.get('kernel_parts_ms_per_step').items()}, r['kernel_launches_per_step'])
    except Exception as e: print('   ??', l[:200])
PY
tail -n 3 gpurun_out/r2p19/b.err
