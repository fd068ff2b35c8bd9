#!/bin/bash
mkdir -p gpurun_out/r2p18; rm -f gpurun_out/r2p18/*
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "128_and_256 or arbitrary or mfma or wide" > gpurun_out/r2p18/pytest.txt 2>&1
tail -n 8 gpurun_out/r2p18/pytest.txt
B="python bench.py --no-cpu-baseline --no-parity"
for v in "--workload cfg3_b128" "--workload cfg3_b128 --opt mfma_tile=256" "--workload cfg3_b128 --opt mfma_ladder=6" "--workload cfg3 --opt mfma_tile=128" "--workload cfg3"; do
  echo "== $v" >> gpurun_out/r2p18/b.jsonl
  timeout 600 $B $v >> gpurun_out/r2p18/b.jsonl 2>> gpurun_out/r2p18/b.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p18/b.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.3f  kernel %.3f ms  %s frac %.4f  achieved %.1f %s' % (d['ms_per_step'], r['kernel_ms_per_step'], r['bound'], r['frac'], r['achieved'], r['unit']), {k: round(v,3) for k,v in r.get('kernel_parts_ms_per_step').items()}, r['kernel_launches_per_step'])
    except Exception as e: print('   ??', l[:200])
PY
tail -n 3 gpurun_out/r2p18/b.err
