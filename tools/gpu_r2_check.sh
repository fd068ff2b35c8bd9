#!/bin/bash
# what the round-end driver runs, plus the cfg5 parity legs
mkdir -p gpurun_out/check; rm -f gpurun_out/check/*
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/check/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/check/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/check/smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/check/smoke.txt
for v in "" "--cfg5-subset 1000" "--cfg5-separate"; do
  timeout 600 python bench.py --workload cfg5 $v > gpurun_out/check/cfg5_$(echo $v | tr -d ' -').json 2> gpurun_out/check/cfg5.err; echo "cfg5 $v rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/check/cfg5_*.json')):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], 'value %.1f' % d['value'], 'frac %.3f' % d['roofline']['frac'], d.get('parity'))
    except Exception as e: print(f, 'BAD', e)
PY
