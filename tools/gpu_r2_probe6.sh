#!/bin/bash
# round 2, probe 6: exact-filter rescoring path: tests + cost
mkdir -p gpurun_out/r2p6
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma or wide or arbitrary or skinny or batch" > gpurun_out/r2p6/pytest.txt 2>&1
tail -n 25 gpurun_out/r2p6/pytest.txt
python bench.py --workload cfg3 --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r2p6/cfg3.json 2> gpurun_out/r2p6/cfg3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2p6/cfg3.json').read()); r=d['roofline']
print('ms/step %.2f kernel %.2f frac %.4f'%(d['ms_per_step'], r['kernel_ms_per_step'], r['frac']), r['other_kernels_ms_per_step'])
print(d.get('parity'))
PY
tail -n 5 gpurun_out/r2p6/cfg3.err
