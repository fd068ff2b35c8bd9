#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/dup2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flagged" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
S="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --rows 1250000 --steps 60 --warmup 10"
F="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --steps 20 --warmup 5"
specs=()
for i in 1 2 3; do specs+=("s_fb$i: $S" "s_nofb$i: $S --opt wide_fallback=0"); done
for i in 1 2; do specs+=("f_fb$i: $F" "f_nofb$i: $F --opt wide_fallback=0"); done
timeout 1700 python tools/bench_variants.py $O "${specs[@]}" 2>&1 | tee $O/variants.txt
