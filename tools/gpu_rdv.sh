#!/bin/bash
# cfg3 with/without tile rendezvous and nt corpus stream
mkdir -p gpurun_out
for opts in "" "--opt mfma_rendezvous=1" "--opt mfma_a_nt=1" "--opt mfma_rendezvous=1 --opt mfma_a_nt=1"; do
  echo "== $opts"
  timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 5 --warmup 2 $opts 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
done 2>&1 | tee gpurun_out/rdv.log
