#!/bin/bash
# A/B of two builds of the library on one box, interleaved: libtavb_base.so (the previous commit) against libtavb.so
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/ablib; mkdir -p $O
F="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --steps 20 --warmup 5"
S="--no-cpu-baseline --no-parity --no-sub --no-calibration --workload cfg3 --rows 1250000 --steps 60 --warmup 10"
specs=()
for i in 1 2 3; do specs+=("f_new$i: $F" "f_base$i: TAVB_LIBRARY=libtavb_base.so $F"); done
for i in 1 2 3; do specs+=("s_new$i: $S" "s_base$i: TAVB_LIBRARY=libtavb_base.so $S"); done
timeout 1700 python tools/bench_variants.py $O "${specs[@]}" 2>&1 | tee $O/variants.txt
