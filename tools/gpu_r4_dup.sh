#!/bin/bash
# round 4: the wide split-plane fallback for batches with MANY flagged queries -- tests, its cost when nothing is flagged, the duplication-cliff workload
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/dup; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "flagged or wide or clustered or mfma or shadow or anisotropic or arbitrary_fp32" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
Q="--no-cpu-baseline --no-parity --no-sub --no-calibration --steps 20 --warmup 5"
timeout 1700 python tools/bench_variants.py $O \
  "cfg3: $Q --workload cfg3" "cfg3_nofb: $Q --workload cfg3 --opt wide_fallback=0" "cfg3_again: $Q --workload cfg3" \
  "shard: $Q --workload cfg3 --rows 1250000 --steps 40" "shard_nofb: $Q --workload cfg3 --rows 1250000 --steps 40 --opt wide_fallback=0" \
  "dup: --no-cpu-baseline --no-sub --no-calibration --steps 5 --warmup 2 --workload cfg3_dup" \
  "dup_nofb: --no-cpu-baseline --no-parity --no-sub --no-calibration --steps 3 --warmup 1 --workload cfg3_dup --opt wide_fallback=0" 2>&1 | tee $O/variants.txt
python - <<'PY'
import json
for n in ("dup","dup_nofb"):
    try:
        d=json.load(open(f"gpurun_out/r4/dup/bench_{n}.json")); print(n, d["ms_per_step"], d.get("flagged_fraction"), d.get("parity"))
    except Exception as e: print(n, e)
PY
