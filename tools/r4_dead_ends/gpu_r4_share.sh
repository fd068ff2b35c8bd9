#!/bin/bash
# round 4: thresholds shared between shards -- correctness tests, then a 1/8 shard of cfg3 on one GPU with seven emulated peers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/share; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "thresholds_shared or one_rank_rccl or ladder or wide_tile or mfma_batch" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
Q="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --rows 1250000 --steps 40 --warmup 10"
timeout 1500 python tools/bench_variants.py $O \
  "shard: $Q" \
  "shard_emul8: $Q --opt share_emulate_world=8" \
  "shard_again: $Q" \
  "shard_emul8_again: $Q --opt share_emulate_world=8" \
  "shard_emul8_parity: --no-cpu-baseline --no-sub --workload cfg3 --rows 1250000 --steps 40 --opt share_emulate_world=8" \
  "shard25_emul4: --no-cpu-baseline --no-parity --no-sub --workload cfg3 --rows 2500000 --steps 20 --opt share_emulate_world=4" \
  "shard25: --no-cpu-baseline --no-parity --no-sub --workload cfg3 --rows 2500000 --steps 20" \
  "cfg4_emul8: --no-cpu-baseline --no-parity --no-sub --workload cfg4 --steps 10 --opt share_emulate_world=8" \
  "cfg4: --no-cpu-baseline --no-parity --no-sub --workload cfg4 --steps 10" 2>&1 | tee $O/variants.txt
