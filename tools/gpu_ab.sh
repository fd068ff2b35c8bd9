#!/bin/bash
# A/B two builds of the library on the same box: new = typeagent_py_amd/libtavb.so, prev = typeagent_py_amd/libtavb_prev.so
# usage: gpu_ab.sh "<bench args>" [rounds]
mkdir -p gpurun_out
cp typeagent_py_amd/libtavb.so /tmp/new.so; cp typeagent_py_amd/libtavb_prev.so /tmp/prev.so
args="$1"; rounds=${2:-2}
for i in $(seq $rounds); do
  for which in new prev; do
    cp /tmp/$which.so typeagent_py_amd/libtavb.so
    echo -n "$which: "
    timeout 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],3), round(r['achieved'],1), round(r['frac'],4), r.get('kernel_avg_ms'))"
  done
done 2>&1 | tee gpurun_out/ab.log
cp /tmp/new.so typeagent_py_amd/libtavb.so
