#!/bin/bash
# Final-state evidence for profiles/: kernel traces of cfg2 / cfg3 / cfg5 and the bench lines.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
timeout 900 python bench.py --steps 200 --warmup 20 > $OUT/final_bench_cfg2.json 2> $OUT/final_bench_cfg2.err; echo "cfg2 rc=$?" | tee -a $OUT/round.log
timeout 900 python bench.py --workload cfg2_f16 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/final_bench_cfg2_f16.json 2> $OUT/final_bench_cfg2_f16.err; echo "cfg2f16 rc=$?" | tee -a $OUT/round.log
timeout 900 python bench.py --workload cfg3 --steps 10 --warmup 2 > $OUT/final_bench_cfg3.json 2> $OUT/final_bench_cfg3.err; echo "cfg3 rc=$?" | tee -a $OUT/round.log
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 3 > $OUT/final_bench_cfg5.json 2> $OUT/final_bench_cfg5.err; echo "cfg5 rc=$?" | tee -a $OUT/round.log
timeout 900 python bench.py --workload cfg5 --steps 20 --warmup 3 --cfg5-separate > $OUT/final_bench_cfg5_separate.json 2> $OUT/final_bench_cfg5_separate.err
timeout 900 python bench.py --workload cfg1 --steps 200 --warmup 20 > $OUT/final_bench_cfg1.json 2> $OUT/final_bench_cfg1.err
timeout 900 python bench.py --workload cfg2_b32 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/final_bench_cfg2_b32.json 2> $OUT/final_bench_cfg2_b32.err
timeout 900 python bench.py --workload cfg2_b1024 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/final_bench_cfg2_b1024.json 2> $OUT/final_bench_cfg2_b1024.err
timeout 900 python tools/batch_sweep.py --dtype fp32 --sizes 1,2,4,5,8,16,32,64,128,256,1024 > $OUT/final_batch_sweep_fp32.jsonl 2>/dev/null
timeout 900 python tools/batch_sweep.py --dtype fp16 --rows 2000000 --sizes 1,2,3,4,8,16,32,33,64,128,256,1024 > $OUT/final_batch_sweep_fp16.jsonl 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_final_cfg2 -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_final_cfg2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_final_cfg3 -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_final_cfg3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_final_cfg5 -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 8 --warmup 2 > $OUT/prof_final_cfg5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_final_cfg2_b1024 -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2_b1024 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_final_cfg2_b1024.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_final_cfg2_b32 -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2_b32 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/prof_final_cfg2_b32.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_final_cfg2_b32_fetch -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2_b32 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_final_cfg2_b32_fetch.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_final_cfg3_fetch -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_final_cfg3_fetch.log 2>&1
cd $GRAFT_REPO_ROOT
for f in final_bench_cfg2 final_bench_cfg2_f16 final_bench_cfg3 final_bench_cfg5 final_bench_cfg5_separate final_bench_cfg1 final_bench_cfg2_b32 final_bench_cfg2_b1024; do echo "--- $f"; cat $OUT/$f.json; done | tee -a $OUT/round.log
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
