#!/bin/bash
# round 3, closing run: the whole GPU suite, the driver's bench invocation, the kernel trace of the headline alone, the 1/8 shard
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out/r3/final
O=$R/gpurun_out/r3/final
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -4 $O/gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? in $(( $(date +%s) - S )) s"; tail -3 $O/bench_default.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_headline -o x -- python $R/bench.py --no-cpu-baseline --no-parity --no-sub --steps 20 --warmup 5 > $O/trace_headline.log 2>&1; echo "trace rc=$?"
cd $R
db=$(find $O/trace_headline -name "*results.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/trace_headline.md
rm -rf $O/trace_headline/
grep -o '"kernel_ms_per_step": [0-9.]*' $O/trace_headline.log | head -2
Q="--no-cpu-baseline --no-parity"
timeout 900 python tools/bench_variants.py $O \
  "shard_1of8: --workload cfg3 --rows 1250000 --steps 40 $Q" \
  "one_rank_dist: TAVB_BENCH_FORCE_DIST=1 --workload cfg3 --rows 1250000 --steps 40 --no-cpu-baseline" 2>&1 | tee $O/variants.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/final/bench_default.json").read().strip().splitlines()[-1])
def show(n, r):
    ro=r.get("roofline",{})
    print(n, "value", round(r.get("value", r.get("queries_per_sec", 0)),1), "ms/step", round(r.get("ms_per_step",0),3), "frac", round(ro.get("frac",0),4), ro.get("bound"), "kern", round(ro.get("kernel_ms_per_step",0),3),
          "parity", (r.get("parity") or {}).get("ok"), (r.get("parity") or {}).get("positions_exact"), (r.get("parity") or {}).get("error"), "flagged", r.get("flagged_fraction"), r.get("vs_gaussian"))
show("cfg3", d)
for n, r in (d.get("sub") or {}).items(): show(n, r)
print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k not in ("sample","thread_sweep_ms","default_threads","one_thread")})
PY
head -8 $O/trace_headline.md | cut -c1-200
