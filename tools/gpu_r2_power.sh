#!/bin/bash
# clocks / power while the cfg3 batch loop runs (is the ~1.6 GHz matrix clock a power cap?)
mkdir -p gpurun_out/r2pow; rm -f gpurun_out/r2pow/smi.txt
python bench.py --workload cfg3 --no-cpu-baseline --no-parity --steps 600 --warmup 2 > gpurun_out/r2pow/bench.json 2> gpurun_out/r2pow/bench.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "GPU\[0\].*(Power|sclk)" | tr '\n' ' ' >> gpurun_out/r2pow/smi.txt
  echo >> gpurun_out/r2pow/smi.txt
  sleep 0.4
done
rocm-smi --showmaxpower 2>/dev/null | grep -E "GPU\[0\]" >> gpurun_out/r2pow/smi.txt
sort -t: -k3 -n gpurun_out/r2pow/smi.txt | tail -12
python -c "
import json; d=json.loads(open('gpurun_out/r2pow/bench.json').read()); print(d['ms_per_step'], d['roofline']['frac'])"
