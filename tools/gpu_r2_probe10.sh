#!/bin/bash
mkdir -p gpurun_out/r2p10
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r2p10/default.json 2> gpurun_out/r2p10/default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2p10/default.json').read().strip().splitlines()[-1])
print('value', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['kernel_parts_ms_per_step'], d['roofline']['other_kernels_ms_per_step'])
print('parity', d['parity'])
print('cpu', d['cpu_baseline'])
for k,v in d['sub'].items():
    print(k, v['queries_per_sec'], v['ms_per_step'], v['roofline']['frac'], v['roofline']['kernel_ms_per_step'], v.get('parity',{}).get('ok'), v.get('parity',{}).get('seconds'), (v.get('cpu_baseline') or {}).get('value'))
PY
tail -n 6 gpurun_out/r2p10/default.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2p10/dist1.json 2> gpurun_out/r2p10/dist1.err
tail -c 1500 gpurun_out/r2p10/dist1.json; tail -n 5 gpurun_out/r2p10/dist1.err
TAVB_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29545 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-sub --scaling weak --rows 2000000 > gpurun_out/r2p10/dist_forced.json 2> gpurun_out/r2p10/dist_forced.err
tail -c 1200 gpurun_out/r2p10/dist_forced.json; tail -n 5 gpurun_out/r2p10/dist_forced.err
