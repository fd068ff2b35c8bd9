#!/bin/bash
# round 3: the whole GPU suite, then the driver's own bench invocation (north-star suite with every sub-record)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r3/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r3/gpu_suite.log
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err
echo "bench rc=$? in $(( $(date +%s) - S )) s"; tail -8 gpurun_out/r3/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/bench_default.json").read().strip().splitlines()[-1])
def show(n, r):
    ro=r.get("roofline",{})
    print(n, "value", round(r.get("value", r.get("queries_per_sec", 0)),1), "ms/step", round(r.get("ms_per_step",0),3), "frac", round(ro.get("frac",0),4), ro.get("bound"),
          "parity", (r.get("parity") or {}).get("ok"), (r.get("parity") or {}).get("positions_exact"), (r.get("parity") or {}).get("error"), "flagged", r.get("flagged_fraction"), r.get("vs_gaussian"))
show("cfg3", d)
for n, r in (d.get("sub") or {}).items(): show(n, r)
print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"})
PY
