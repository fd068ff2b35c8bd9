#!/bin/bash
# round 4: admission path v3 (wave masks in SGPRs) -- tests, then A/B against round 3's path (mfma_sched=6) on the full corpus and on a 1/8 shard
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/slow2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "mfma or wide or clustered or anisotropic or shadow or arbitrary_fp32 or 128_and_256 or thresholds_shared" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
Q="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 20 --warmup 5"
S="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --rows 1250000 --steps 40 --warmup 10"
timeout 1500 python tools/bench_variants.py $O \
  "new1: $Q" "old1: $Q --opt mfma_sched=6" "new2: $Q" "old2: $Q --opt mfma_sched=6" "new3: $Q" "old3: $Q --opt mfma_sched=6" \
  "shard_new1: $S" "shard_old1: $S --opt mfma_sched=6" "shard_new2: $S" "shard_old2: $S --opt mfma_sched=6" \
  "new_parity: --no-cpu-baseline --no-sub --workload cfg3 --steps 10" 2>&1 | tee $O/variants.txt
