#!/bin/bash
# One gpurun call: GPU parity suite, first bench line, geometry sweep, rocprof kernel trace.
# Everything is bounded by its own timeout; logs go to gpurun_out/ (merged back by gpurun).
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== box: $(hostname) $(date -u +%FT%TZ)" | tee $OUT/round.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 | tee -a $OUT/round.log
nproc | tee -a $OUT/round.log
free -g | head -2 | tee -a $OUT/round.log

echo "== pytest gpu" | tee -a $OUT/round.log
timeout 900 python -m pytest tests -m gpu -q -n 1 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round.log
tail -5 $OUT/pytest_gpu.log | tee -a $OUT/round.log

echo "== smoke" | tee -a $OUT/round.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" | tee -a $OUT/round.log
tail -3 $OUT/smoke.log | tee -a $OUT/round.log

echo "== bench" | tee -a $OUT/round.log
timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
echo "bench rc=$?" | tee -a $OUT/round.log
cat $OUT/bench_cfg2.json | tee -a $OUT/round.log
tail -5 $OUT/bench_cfg2.err | tee -a $OUT/round.log

echo "== sweep" | tee -a $OUT/round.log
timeout 600 python tools/sweep_scan.py --tag f32 > $OUT/sweep_f32.log 2>&1
echo "sweep rc=$?" | tee -a $OUT/round.log
tail -2 $OUT/sweep_f32.log | tee -a $OUT/round.log

echo "== rocprof kernel trace" | tee -a $OUT/round.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_cfg2 -o cfg2 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_cfg2.log 2>&1
echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/$OUT/round.log
cd $GRAFT_REPO_ROOT
find $OUT/prof_cfg2 -name "*stats*" | head | tee -a $OUT/round.log
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
