#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "== $(date -u +%FT%TZ)" | tee $OUT/round.log
timeout 600 python -m pytest tests -m gpu -q -n 1 --tb=short -p no:cacheprovider -k "sharded or device_only" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round.log
tail -8 $OUT/pytest_gpu.log | tee -a $OUT/round.log

echo "== torchrun 1-rank dry run of the N>1 bench path" | tee -a $OUT/round.log
TAVB_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err
echo "rc=$?" | tee -a $OUT/round.log
cat $OUT/bench_dist1.json | tee -a $OUT/round.log
tail -5 $OUT/bench_dist1.err | tee -a $OUT/round.log

echo "== counters available" | tee -a $OUT/round.log
rocprofv3 -L 2>/dev/null | grep -i -E "^\s*(Name|Counter)?.*(MFMA|FETCH_SIZE|WRITE_SIZE|LDS_BANK_CONFLICT|TCC_HIT|TCC_MISS|TCC_EA0_RDREQ|GRBM_GUI_ACTIVE|SQ_WAVES|SQ_BUSY_CYCLES|MfmaUtil|VALUBusy)" | head -60 > $OUT/counters.txt
wc -l $OUT/counters.txt | tee -a $OUT/round.log

cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C cfg2" | tee -a $OUT/round.log
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_cfg2_$C -o cfg2 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/pmc_cfg2_$C.log 2>&1
  echo "rc=$?" | tee -a $OUT/round.log
done
for C in FETCH_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  TAG=$(echo $C | tr ' ' '_' | cut -c1-40)
  echo "== pmc $TAG cfg3 (2M rows)" | tee -a $OUT/round.log
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_cfg3_$TAG -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --rows 2000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_cfg3_$TAG.log 2>&1
  echo "rc=$?" | tee -a $OUT/round.log
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*counter_collection*" | head -20 | tee -a $OUT/round.log
echo "== done $(date -u +%FT%TZ)" | tee -a $OUT/round.log
