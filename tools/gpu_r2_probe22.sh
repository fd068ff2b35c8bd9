#!/bin/bash
mkdir -p gpurun_out/r2p22; rm -f gpurun_out/r2p22/*
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32_shadow or f32_corpus" > gpurun_out/r2p22/pytest.txt 2>&1
tail -n 12 gpurun_out/r2p22/pytest.txt
B="python bench.py --no-cpu-baseline"
for v in "--workload cfg2 --opt f32_shadow=2" "--workload cfg2 --no-parity" "--workload cfg2_b32 --opt f32_shadow=2" "--workload cfg2_b32 --no-parity" "--workload cfg2_b32 --queries 8 --opt f32_shadow=2" "--workload cfg2_b32 --queries 8 --no-parity" "--workload cfg2_b32 --queries 64 --opt f32_shadow=2 --no-parity" "--workload cfg1 --opt f32_shadow=2 --no-parity"; do
  echo "== $v" >> gpurun_out/r2p22/b.jsonl
  timeout 600 $B $v >> gpurun_out/r2p22/b.jsonl 2>> gpurun_out/r2p22/b.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p22/b.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.3f  kernel %.3f ms  %s frac %.4f  q/s %.0f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['bound'], r['frac'], d['value']), {k: round(v,3) for k,v in r.get('kernel_parts_ms_per_step').items()}, 'other', {k: round(v,3) for k,v in r.get('other_kernels_ms_per_step',{}).items()}, 'shadow' if 'scanned' in r else '', 'parity', (d.get('parity') or {}).get('ok'), (d.get('parity') or {}).get('positions_exact'))
    except Exception as e: print('   ??', l[:200])
PY
tail -n 3 gpurun_out/r2p22/b.err
