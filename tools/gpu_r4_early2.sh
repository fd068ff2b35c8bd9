#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/early; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "doomed or flagged or profile_counters" > $O/tests2.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests2.log
D="--no-cpu-baseline --no-sub --no-calibration --workload cfg3_dup --steps 10 --warmup 3"
timeout 1700 python tools/bench_variants.py $O "dup_early: $D" "dup_late: $D --opt early_exact=0" 2>&1 | tee $O/variants2.txt
python - <<'PY'
import json
for n in ("dup_early","dup_late"):
    d=json.load(open(f"gpurun_out/r4/early/bench_{n}.json"))
    print(n, d["ms_per_step"], d.get("flagged_fraction"), json.dumps(d["roofline"])[:700])
PY
