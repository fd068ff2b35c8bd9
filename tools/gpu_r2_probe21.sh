#!/bin/bash
mkdir -p gpurun_out/r2p21; rm -f gpurun_out/r2p21/*
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multidevice.py -x -q -m gpu -k "f32_corpus_large or f32_shadow or skinny_kernel_serves or device_group_matches or batch_equals or rerank or fused" > gpurun_out/r2p21/pytest.txt 2>&1
tail -n 12 gpurun_out/r2p21/pytest.txt
B="python bench.py --no-cpu-baseline"
for v in "--workload cfg2_b1024" "--workload cfg2_b1024 --opt f32_shadow=0 --no-parity" "--workload cfg2_b1024 --queries 128" "--workload cfg2_b1024 --queries 128 --opt f32_shadow=0 --no-parity" "--workload cfg2_b1024 --queries 256" "--workload cfg2_b1024 --queries 64 --no-parity"; do
  echo "== $v" >> gpurun_out/r2p21/b.jsonl
  timeout 600 $B $v >> gpurun_out/r2p21/b.jsonl 2>> gpurun_out/r2p21/b.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p21/b.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.3f  kernel %.3f ms  %s frac %.4f of %s  q/s %.0f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['bound'], r['frac'], r['peak'], d['value']), {k: round(v,3) for k,v in r.get('kernel_parts_ms_per_step').items()}, 'other', {k: round(v,3) for k,v in r.get('other_kernels_ms_per_step',{}).items()}, 'parity', (d.get('parity') or {}).get('ok'), (d.get('parity') or {}).get('positions_exact'))
    except Exception as e: print('   ??', l[:200])
PY
tail -n 3 gpurun_out/r2p21/b.err
