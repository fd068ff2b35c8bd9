#!/bin/bash
# round 3: the template ablations of mfma_scan_kernel again, on the shipping (band selection) kernel, one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
Q="--workload cfg3 --no-cpu-baseline --no-parity --steps 10"
timeout 1500 python tools/bench_variants.py gpurun_out/r3/ablate \
  "mode0_everything: $Q" \
  "mode256_no_admissions: $Q --opt mfma_ablate=256" \
  "mode264_query_resident: $Q --opt mfma_ablate=264" \
  "mode260_corpus_l2: $Q --opt mfma_ablate=260" \
  "mode268_both_resident: $Q --opt mfma_ablate=268" \
  "mode258_no_dma: $Q --opt mfma_ablate=258" \
  "mode0_again: $Q" \
  "sched1: $Q --opt mfma_sched=1" \
  "sched2: $Q --opt mfma_sched=2" 2>&1 | tee gpurun_out/r3/ablate_variants.txt
