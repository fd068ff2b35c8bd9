#!/bin/bash
mkdir -p gpurun_out/r2p13
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma or wide or arbitrary or batch or ladder or cfg3" > gpurun_out/r2p13/pytest.txt 2>&1
tail -n 6 gpurun_out/r2p13/pytest.txt
B="python bench.py --workload cfg3 --no-cpu-baseline --no-parity --steps 10 --warmup 3"
for v in "--rows 131072 --opt mfma_variant=3" "--rows 131072 --opt mfma_variant=6" "--rows 32768 --opt mfma_variant=6" "--rows 40960 --opt mfma_variant=6" "--rows 1250000" "--rows 1250000 --opt mfma_sample_rows=40960" "--rows 1250000 --opt mfma_sample_rows=40960 --opt mfma_v6_min_rows=0" \
  "" "--opt mfma_sample_rows=40960" "--opt mfma_sample_rows=40960 --opt mfma_v6_min_rows=0" "--opt mfma_sample_rows=40960 --opt mfma_v6_min_rows=0 --opt mfma_ladder=5" ""; do
  echo "== $v" >> gpurun_out/r2p13/cfg3.jsonl
  $B $v >> gpurun_out/r2p13/cfg3.jsonl 2>> gpurun_out/r2p13/cfg3.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p13/cfg3.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.3f  kernel %.3f ms  frac %.4f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac']), {k: round(v,3) for k,v in r.get('kernel_parts_ms_per_step').items()}, r['kernel_launches_per_step'], {k: round(v,3) for k,v in r['other_kernels_ms_per_step'].items()})
    except Exception as e: print('   ??', l[:200])
PY
tail -n 3 gpurun_out/r2p13/cfg3.err
