#!/bin/bash
# NOTE: kept as run (mid round 2).  The `mfma_variant` option no longer exists -- variant 6 is the only 256-query kernel --, so drop it to re-run;
# variant 5 (round 1's 4-wave 384 x 256 tile) was deleted after this measurement.
# round 2, probe 2: variant 6 (K = 64 whole-line staging) correctness + speed
mkdir -p gpurun_out/r2p2
true
true
B="python bench.py --workload cfg3 --no-cpu-baseline --no-parity --steps 5 --warmup 2"
for v in "mfma_variant=6" "mfma_variant=6 --opt mfma_sched=1" "mfma_variant=6 --opt mfma_sched=2" "mfma_variant=6 --opt mfma_sched=3" \
         "mfma_variant=6 --opt mfma_ablate=256" "mfma_variant=6 --opt mfma_ablate=260" "mfma_variant=6 --opt mfma_ablate=258" "mfma_variant=6 --opt mfma_ablate=512" \
         "mfma_variant=5" "mfma_variant=5 --opt mfma_ablate=256" "mfma_variant=0"; do
  echo "== $v" >> gpurun_out/r2p2/cfg3.jsonl
  $B --opt $v >> gpurun_out/r2p2/cfg3.jsonl 2>> gpurun_out/r2p2/cfg3.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2p2/cfg3.jsonl'):
    if l.startswith('=='): print(l.strip()); continue
    try:
        d=json.loads(l); r=d['roofline']; print('   ms/step %.2f  kernel %.2f ms  frac %.4f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac']), r.get('other_kernels_ms_per_step'))
    except Exception as e: print('   ??', l[:200])
PY
