#!/bin/bash
mkdir -p gpurun_out/r2p11
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma or wide or arbitrary or skinny or batch or ladder" > gpurun_out/r2p11/pytest.txt 2>&1
tail -n 6 gpurun_out/r2p11/pytest.txt
B="python bench.py --workload cfg3 --no-cpu-baseline --no-parity --steps 5 --warmup 2"
for v in "mfma_variant=0" "mfma_variant=6" "mfma_variant=0 --opt mfma_sample_rows=32768" "mfma_variant=6 --opt mfma_sample_rows=32768" "mfma_variant=6 --opt mfma_sample_rows=8192" "mfma_variant=0"; do
  echo "== $v" >> gpurun_out/r2p11/cfg3.jsonl
  $B --opt $v >> gpurun_out/r2p11/cfg3.jsonl 2>> gpurun_out/r2p11/cfg3.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p11/cfg3.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.2f  kernel %.2f ms  frac %.4f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac']), {k: round(v,2) for k,v in r.get('kernel_parts_ms_per_step').items()}, r['kernel_launches_per_step'])
    except Exception as e: print('   ??', l[:200])
PY
