#!/bin/bash
# round 4: branch-free admission path of the wide tile -- tests first, then timing next to the stall ablations (284: no wait for the staging loads,
# 332: query pieces only, 396: corpus pieces only; all on top of 268 = both operands cache resident, no admissions)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/slow; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "mfma or wide or clustered or anisotropic or shadow or arbitrary_fp32 or 128_and_256" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
Q="--no-cpu-baseline --no-parity --no-sub --workload cfg3 --steps 20 --warmup 5"
timeout 1500 python tools/bench_variants.py $O \
  "ship: $Q" \
  "abl256: $Q --opt mfma_ablate=256" \
  "abl268: $Q --opt mfma_ablate=268" \
  "abl284_nowait: $Q --opt mfma_ablate=284" \
  "abl332_queries_only: $Q --opt mfma_ablate=332" \
  "abl396_corpus_only: $Q --opt mfma_ablate=396" \
  "abl258: $Q --opt mfma_ablate=258" \
  "ship_again: $Q" 2>&1 | tee $O/variants.txt
