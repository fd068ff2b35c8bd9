#!/bin/bash
# round 4: ladder shapes again, with the new admission path (dense first phases now pay one LDS atomic per admitted row)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/ladder2; mkdir -p $O
S="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --rows 1250000 --steps 40 --warmup 10"
F="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 20 --warmup 5"
timeout 1700 python tools/bench_variants.py $O \
  "s20480x4: $S" "s10240x4: $S --opt mfma_sample_rows=10240" "s5120x4: $S --opt mfma_sample_rows=5120" \
  "s10240x8: $S --opt mfma_sample_rows=10240 --opt mfma_ladder=8" "s5120x8: $S --opt mfma_sample_rows=5120 --opt mfma_ladder=8" "s10240x6: $S --opt mfma_sample_rows=10240 --opt mfma_ladder=6" \
  "s20480x4_again: $S" \
  "f20480x4: $F" "f10240x4: $F --opt mfma_sample_rows=10240" "f5120x8: $F --opt mfma_sample_rows=5120 --opt mfma_ladder=8" "f10240x8: $F --opt mfma_sample_rows=10240 --opt mfma_ladder=8" "f10240x6: $F --opt mfma_sample_rows=10240 --opt mfma_ladder=6" "f20480x4_again: $F" 2>&1 | tee $O/variants.txt
