#!/bin/bash
# what the round-end driver runs: the GPU suite, smoke(), the default bench line (+ a one-rank torchrun launch of the sharded path)
mkdir -p gpurun_out/check; rm -f gpurun_out/check/*
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/check/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/check/pytest.txt | head -3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/check/smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 1 gpurun_out/check/smoke.txt
( time python bench.py ) > gpurun_out/check/bench_default.json 2> gpurun_out/check/bench_default.err; echo "bench rc=$?"; grep real gpurun_out/check/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/check/bench_torchrun1.json 2> gpurun_out/check/bench_torchrun1.err; echo "torchrun rc=$?"
python - <<'PY'
import json
for f in ('bench_default','bench_torchrun1'):
    try:
        raw=open(f'gpurun_out/check/{f}.json').read()
        d=json.loads(raw); r=d['roofline']
        print(f, 'lines', len(raw.strip().splitlines()), 'value %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'frac %.4f' % r['frac'], 'traffic', r.get('traffic'), 'parity', d['parity']['ok'], d['parity']['positions_exact'], 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'n_gpus', d['n_gpus'], d['scaling'])
        for k,v in (d.get('sub') or {}).items(): print('   ', k, round(v['queries_per_sec'],1), round(v['roofline']['frac'],4), v['parity']['ok'], v['roofline'].get('traffic'))
    except Exception as e: print(f, 'BAD', e)
PY
