#!/bin/bash
# round 4: the early verdict on a batch of doomed bands (early_exact) -- tests, cfg3_dup with and without, cfg3 / clustered / shard for regressions
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/early; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "doomed or flagged or band or wide_tile or cfg3" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
D="--no-cpu-baseline --no-sub --no-calibration --workload cfg3_dup --steps 10 --warmup 3"
F="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --steps 20 --warmup 5"
C="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3_clustered --steps 20 --warmup 5"
S="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --rows 1250000 --steps 60 --warmup 10"
timeout 1700 python tools/bench_variants.py $O "dup_early: $D" "dup_late: $D --opt early_exact=0" "full: $F" "full_off: $F --opt early_exact=0" "clustered: $C" "shard: $S" "shard_off: $S --opt early_exact=0" "dup_early2: $D --no-parity" 2>&1 | tee $O/variants.txt
