#!/bin/bash
# round 4: the driver's invocation of bench.py, line size and the fields the judge asked for
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/bench; mkdir -p $O
S=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? in $(( $(date +%s) - S )) s, line bytes: $(wc -c < $O/bench_default.json)"; tail -5 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/bench/bench_default.json").read().strip().splitlines()[-1])
def show(n, r):
    ro=r.get("roofline",{}); pa=r.get("parity") or {}
    print(n, "value", round(r.get("value", r.get("queries_per_sec", 0)),1), "ms/step", round(r.get("ms_per_step",0),4), "frac", round(ro.get("frac",0),4), ro.get("bound"), "kern", round(ro.get("kernel_ms_per_step",0),3),
          "parity", pa.get("ok"), pa.get("positions_exact"), pa.get("positions_permuted"), "gap", pa.get("max_permuted_gap"), "inv", pa.get("gpu_inversions_vs_f64"), pa.get("reference_inversions_vs_f64"), "noise", pa.get("noise_gpu"), pa.get("noise_ref"), pa.get("error"), "flagged", r.get("flagged_fraction"), r.get("class_api"))
show("cfg3", d); print("sustained", d["roofline"].get("sustained"))
for n, r in (d.get("sub") or {}).items(): show(n, r)
print("cpu", d.get("cpu_baseline"))
PY
