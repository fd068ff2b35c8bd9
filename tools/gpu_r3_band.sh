#!/bin/bash
# round 3, first light of band selection: the wide-tile tests, then cfg3 on gaussian and clustered data
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_multidevice.py -x -q -m gpu \
  -k "wide_tile or mfma or clustered or arbitrary_fp32 or 128_and_256 or f32_corpus_large or f32_shadow or library_loaded or rccl or two_ranks or in_place_edit" \
  > gpurun_out/r3/band_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3/band_tests.log
tail -15 gpurun_out/r3/band_tests.log
timeout 400 python bench.py --workload cfg3 --no-cpu-baseline > gpurun_out/r3/bench_cfg3_band.json 2> gpurun_out/r3/bench_cfg3_band.err
echo "cfg3 rc=$?"; tail -3 gpurun_out/r3/bench_cfg3_band.err
timeout 400 python bench.py --workload cfg3_clustered --no-cpu-baseline > gpurun_out/r3/bench_cfg3_clustered.json 2> gpurun_out/r3/bench_cfg3_clustered.err
echo "clustered rc=$?"; tail -3 gpurun_out/r3/bench_cfg3_clustered.err
TAVB_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --rows 2000000 --steps 5 > gpurun_out/r3/bench_cfg3_one_rank_dist.json 2> gpurun_out/r3/bench_cfg3_one_rank_dist.err
echo "one-rank dist rc=$?"; tail -3 gpurun_out/r3/bench_cfg3_one_rank_dist.err
python - <<'PY'
import json
for n in ("cfg3_band","cfg3_clustered","cfg3_one_rank_dist"):
    try:
        d=json.loads(open(f"gpurun_out/r3/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "qps",round(d["value"]), "ms",round(d["ms_per_step"],2), "frac",round(d["roofline"]["frac"],4), "kern_ms",round(d["roofline"]["kernel_ms_per_step"],2),
              "other",d["roofline"]["other_kernels_ms_per_step"], "flagged",d.get("flagged_queries_per_batch"),"parity",d.get("parity",{}).get("ok"), d.get("parity",{}).get("positions_exact"), d.get("parity",{}).get("error"))
    except Exception as e:
        print(n,"failed",e)
PY
