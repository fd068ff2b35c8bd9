#!/bin/bash
# round 4: the one-launch path for a FEW queries (2 .. 8) and the bigger list budget -- tests, small-batch latencies, single lookups again
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/few; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_corpus or lookup_texts or batched or profile_counters or differential" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python tools/small_batch_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/small_batches_after.txt
for spec in "10000 50 0.85"; do
  timeout 300 python tools/latency_breakdown.py $spec > $O/latency_$(echo $spec | tr ' ' '_').txt 2>&1; grep -E "^rows|ONE launch|scan_blocks|two launches|fuzzy_lookup" $O/latency_$(echo $spec | tr ' ' '_').txt
done
