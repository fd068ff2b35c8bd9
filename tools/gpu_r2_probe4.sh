#!/bin/bash
# round 2, probe 4: L2 touch-ahead distances, ladder geometries, correctness of the touch form
mkdir -p gpurun_out/r2p4
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma" > gpurun_out/r2p4/pytest_mfma.txt 2>&1
tail -n 3 gpurun_out/r2p4/pytest_mfma.txt
B="python bench.py --workload cfg3 --no-cpu-baseline --no-parity --steps 5 --warmup 2"
for v in "mfma_variant=6 --opt mfma_sched=1" "mfma_variant=6 --opt mfma_sched=2" "mfma_variant=6" "mfma_variant=6 --opt mfma_sched=3" \
         "mfma_variant=6 --opt mfma_ablate=256" "mfma_variant=6 --opt mfma_ablate=257" \
         "mfma_variant=0" "mfma_variant=0 --opt mfma_sample_rows=32768 --opt mfma_ladder=19" "mfma_variant=0 --opt mfma_sample_rows=65536 --opt mfma_ladder=9" \
         "mfma_variant=0 --opt mfma_sample_rows=16384 --opt mfma_ladder=7"; do
  echo "== $v" >> gpurun_out/r2p4/cfg3.jsonl
  $B --opt $v >> gpurun_out/r2p4/cfg3.jsonl 2>> gpurun_out/r2p4/cfg3.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p4/cfg3.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.2f  kernel %.2f ms  frac %.4f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac']), {k: round(v,2) for k,v in r.get('kernel_parts_ms_per_step').items()}, r['kernel_launches_per_step'])
    except Exception as e: print('   ??', l[:200])
PY
tail -n 3 gpurun_out/r2p4/cfg3.err
