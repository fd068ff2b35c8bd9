#!/bin/bash
# The one GPU-box script (run through gpurun).  Usage:  bash tools/gpu.sh LABEL STEP [STEP ...]
# Output -> gpurun_out/LABEL/.  Steps:
#   suite            the whole `-m gpu` test suite (-x)                          -> gpu_suite.log
#   tests:EXPR       pytest -m gpu -k EXPR                                       -> gpu_tests.log
#   opttests:OPTS:EXPR  the same under TAVB_ENGINE_OPTIONS=OPTS (every engine of the test process gets those options)   -> gpu_opttests.log
#   fuzz:BASES       tests/test_gpu_fuzz.py once per seed base of the comma-separated list (TAVB_FUZZ_BASE)             -> fuzz_<base>.log
#   smoke            __graft_entry__.smoke()
#   bench            the driver's command (bench.py --gpus 1 --steps 20 --warmup 5) + a digest of the line  -> bench_default.json
#   dist1            the N > 1 code path on a one-rank communicator (TAVB_BENCH_FORCE_DIST=1)               -> bench_dist1.json
#   dist2[:ROWS], dist8[:ROWS]  the N = 2 (N = 8) branches of bench.py on ONE GPU: the ranks under torch.distributed.run share cuda:0, gloo, the lookups' exchange
#                    through the host (TAVB_BENCH_DIST_BACKEND=gloo); ROWS rows in the strong-scaling corpus and per rank of cfg4_weak (default 2000000)  -> bench_dist2.json
#   parity100m       how long rank 0's whole-corpus parity pass takes at N = 8 (100M rows: 100 chunks generated on the device, 16 queries): tools/parity_leg_time.py -> parity100m.txt
#   toolcheck        every measurement script under tools/ once, with small shapes, against the library that ships: rc per script -> toolcheck.txt
#                    (tools/README.md's "runs against ABI 6" column comes from here)
#   sweep            tools/regime_sweep.py: corpus size x batch size per dtype, every cell slower than a cell with more rows AND queries -> regime_sweep.md
#   timeline         one steady-state lookup launch by launch (rocprofv3 kernel trace: 1.25M-row shard, cfg2_b32)    -> timeline_*.md
#   variants:FILE    bench.py variants listed one per line ("name: args") in FILE, one child process each   -> bench_<name>.json, variants.txt
#   ab:ARGS          interleaved A/B on this box: A = cfg3 as shipped, B = the same + ARGS (3 rounds)
#   pmc              FETCH_SIZE passes for every workload of profiles/pmc_traffic.json + the MFMA counters of cfg3 -> pmc_*.md, pmc_traffic.json
#   pmcw             WRITE_SIZE pass of cfg3
#   trace            rocprofv3 --kernel-trace --stats of the driver's command (fewer steps) and of the headline alone -> trace_*.md
#   power            rocm-smi power / sclk samples under a 400-step cfg3 loop    -> smi.txt
#   ceiling          tools/ceiling.py: shipping lookup / MFMA-only ablation / vendor GEMM on the cfg3 contraction, each with power and clock -> ceiling.md
#   asan             the C ABI under the ASan/UBSan build of the host side (make -C typeagent_py_amd/csrc debug first; the driver is the
#                    torch-free tools/asan_exercise.py: torch's CUDA init does not survive the ASan runtime)          -> asan.txt
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
LABEL=$1; shift
O=$R/gpurun_out/$LABEL; mkdir -p "$O"
Q="--no-cpu-baseline --no-parity --no-sub --no-calibration"

digest() { # file
python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(n, r):
    ro = r.get("roofline", {}); pa = r.get("parity") or {}
    print(f"{n:16s} value {r.get('value', r.get('queries_per_sec', 0)):10.1f} ms/step {r.get('ms_per_step', 0):8.4f} frac {ro.get('frac', 0):.4f} {ro.get('bound')} "
          f"kern {ro.get('kernel_ms_per_step', 0) or 0:.3f} parity {pa.get('ok')} {pa.get('positions_exact')}/{pa.get('positions_permuted')} inv {pa.get('gpu_inversions_vs_f64')}/{pa.get('reference_inversions_vs_f64')} "
          f"noise {pa.get('noise_gpu')}/{pa.get('noise_ref')} {pa.get('error') or ''} flagged {r.get('flagged_fraction')} {r.get('class_api') or ''} {r.get('variants') or ''}")
show("headline", d); print("sustained", d["roofline"].get("sustained")); print("cpu", d.get("cpu_baseline"))
for n, r in (d.get("sub") or {}).items(): show(n, r)
PY
}

pmc() { # name pmc_traffic-key|"" workload-args counters...
  local name=$1 key=$2 wl="$3"; shift 3
  ( cd /tmp && timeout -k 5 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/raw_$name -- python $R/bench.py $Q $wl > $O/pmc_$name.log 2>&1 )
  echo "pmc $name rc=$? $(date -u +%T)"
  local total=$(grep -o "[0-9]* lookups in this process" $O/pmc_$name.log | head -1 | cut -d" " -f1)
  python tools/pmc_summary.py $O/raw_$name/*/*_counter_collection.csv --steps ${total:-1} --cmd "bench.py $Q $wl" ${key:+--json $O/pmc_traffic.json --name $key} > $O/pmc_$name.md 2>> $O/pmc_err.log
  rm -rf $O/raw_$name
}

for step in "$@"; do
  arg=${step#*:}; S=$(date +%s)
  case ${step%%:*} in
    suite) timeout 1700 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.log ;;
    tests) timeout 1200 python -m pytest tests -q -m gpu -x -k "$arg" > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/gpu_tests.log ;;
    opttests) TAVB_ENGINE_OPTIONS="${arg%%:*}" timeout 1200 python -m pytest tests -q -m gpu -x -k "${arg#*:}" > $O/gpu_opttests.log 2>&1; echo "opttests(${arg%%:*}) rc=$?"; tail -6 $O/gpu_opttests.log ;;
    fuzz) for b in ${arg//,/ }; do TAVB_FUZZ_BASE=$b timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu > $O/fuzz_$b.log 2>&1; echo "fuzz base $b rc=$? $(tail -1 $O/fuzz_$b.log)"; done ;;
    smoke) timeout 180 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)
      timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
      echo "bench rc=$? in $(( $(date +%s) - S )) s, line bytes: $(wc -c < $O/bench_default.json)"; tail -3 $O/bench_default.err; digest $O/bench_default.json ;;
    dist1)
      TAVB_BENCH_FORCE_DIST=1 timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err
      echo "dist1 rc=$? line bytes: $(wc -c < $O/bench_dist1.json)"; tail -3 $O/bench_dist1.err; digest $O/bench_dist1.json ;;
    dist2|dist8)
      NP=${step%%:*}; NP=${NP#dist}
      ROWS=2000000; [ "$arg" != "dist$NP" ] && ROWS=$arg
      PORT=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
      TAVB_BENCH_DIST_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NP --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $NP --steps 5 --warmup 2 --rows $ROWS > $O/bench_dist$NP.json 2> $O/bench_dist$NP.err
      echo "dist$NP rc=$? line bytes: $(wc -c < $O/bench_dist$NP.json)"; grep -v "^W0\|^\*\*\*\|OMP_NUM\|Gloo\|socket.cpp" $O/bench_dist$NP.err | tail -5; digest $O/bench_dist$NP.json
      python -c "import json; d=json.loads(open('$O/bench_dist$NP.json').read().strip().splitlines()[-1]); print('exchange', d.get('exchange')); print('weak', d['sub']['cfg4_weak'].get('exchange'), d['sub']['cfg4_weak'].get('parity')); print(d.get('dry_run'))" ;;
    parity100m) timeout 1700 python tools/parity_leg_time.py > $O/parity100m.txt 2>&1; echo "parity100m rc=$?"; tail -5 $O/parity100m.txt ;;
    toolcheck)
      : > $O/toolcheck.txt
      tc() { local name=$1; shift; local t0=$(date +%s); timeout 300 "$@" > $O/tool_$name.log 2>&1; echo "$name rc=$? $(( $(date +%s) - t0 )) s" | tee -a $O/toolcheck.txt; }
      tc batch_sweep python tools/batch_sweep.py --rows 200000 --sizes 1,8,32,128
      tc gemm_rate python tools/gemm_rate.py
      tc host_submit_time python tools/host_submit_time.py 200000 1536 fp16 32
      tc inline_query_probe python tools/inline_query_probe.py
      tc latency_breakdown python tools/latency_breakdown.py 10000 10 0.0
      tc load_bench python tools/load_bench.py
      tc oracle_noise python tools/oracle_noise.py
      tc small_batch_latency python tools/small_batch_latency.py
      tc small_scan_forms python tools/small_scan_forms.py
      tc sweep_scan python tools/sweep_scan.py --rows 200000 --quick --tag toolcheck
      tc ceiling python tools/ceiling.py --seconds 2 --rows 2000000
      tc parity_leg_time python tools/parity_leg_time.py --world 2 --rows-per-rank 1000000
      tc regime_sweep python tools/regime_sweep.py --dtype fp16 --rows 1000,50000 --sizes 1,32,64,65,256,1024
      tc mantissa_power python tools/mantissa_power.py --seconds 1 --rows 1000000 --bits 10,0 --corpus-bits 0
      tc group_sweep python tools/group_sweep.py --rows 1000,5000 --sizes 4,32,100 --k 10 --forms default,off,2 --kernel
      tc class_sweep python tools/class_sweep.py --rows 1000,10000 --sizes 1,16,64
      tc terms_breakdown python tools/terms_breakdown.py
      tc misc_sweep python tools/misc_sweep.py --rows 300000
      tc timeline bash tools/timeline.sh
      for mb in load_paths issue_cost kernarg_query flag_completion; do
        tc mb_$mb bash -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_$mb tools/microbench/$mb.hip && /tmp/mb_$mb"
      done ;;
    sweep) timeout 900 python tools/regime_sweep.py > $O/regime_sweep.md 2> $O/regime_sweep.err; echo "sweep rc=$?"; grep -A3 "^cells slower" $O/regime_sweep.md | cut -c1-160 ;;
    timeline) bash tools/timeline.sh; cp $R/gpurun_out/r6z/timeline_*.md $O/ 2>/dev/null; head -6 $O/timeline_shard.md ;;
    variants) mapfile -t specs < <(grep -v '^#' "$arg" | grep .); timeout 1700 python tools/bench_variants.py $O "${specs[@]}" 2>&1 | tee -a $O/variants.txt ;;
    ab)
      specs=(); for i in 1 2 3; do specs+=("A$i: $Q --workload cfg3 --steps 20 --warmup 5" "B$i: $Q --workload cfg3 --steps 20 --warmup 5 $arg"); done
      timeout 1700 python tools/bench_variants.py $O "${specs[@]}" 2>&1 | tee -a $O/variants.txt ;;
    pmc)
      rm -f $O/pmc_traffic.json
      pmc cfg3_fetch cfg3 "--workload cfg3 --steps 2 --warmup 1" FETCH_SIZE
      pmc cfg3_mfma "" "--workload cfg3 --steps 2 --warmup 1" GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
      pmc cfg3_clustered_fetch cfg3_clustered "--workload cfg3_clustered --steps 2 --warmup 1" FETCH_SIZE
      pmc cfg3_q1_fetch cfg3_q1 "--workload cfg3_q1 --steps 5 --warmup 1" FETCH_SIZE
      pmc cfg2_fetch cfg2 "--workload cfg2 --steps 10 --warmup 2" FETCH_SIZE
      pmc cfg4_fetch cfg4 "--workload cfg4 --steps 2 --warmup 1" FETCH_SIZE
      pmc cfg3_b128_fetch cfg3_b128 "--workload cfg3_b128 --steps 2 --warmup 1" FETCH_SIZE
      pmc cfg2_b32_fetch cfg2_b32 "--workload cfg2_b32 --steps 5 --warmup 1" FETCH_SIZE
      pmc cfg3_b32_fetch cfg3_b32 "--workload cfg3_b32 --steps 2 --warmup 1" FETCH_SIZE
      pmc cfg3_subset_fetch cfg3_subset "--workload cfg3_subset --steps 10 --warmup 2" FETCH_SIZE
      pmc cfg2_d3072_fetch cfg2_d3072 "--workload cfg2_d3072 --steps 10 --warmup 2" FETCH_SIZE
      pmc cfg3_d3072_q1_fetch cfg3_d3072_q1 "--workload cfg3_d3072_q1 --steps 5 --warmup 1" FETCH_SIZE
      pmc cfg3_d3072_fetch cfg3_d3072 "--workload cfg3_d3072 --steps 2 --warmup 1" FETCH_SIZE
      pmc cfg3_aniso_fetch cfg3_aniso "--workload cfg3_aniso --steps 2 --warmup 1" FETCH_SIZE
      python -c "import json; d=json.load(open('$O/pmc_traffic.json')); print({k:(round(v['traffic_bytes_per_step']/1e9,3), v['launches_per_step']) for k,v in d.items()})" ;;
    pmcw) pmc cfg3_write "" "--workload cfg3 --steps 2 --warmup 1" WRITE_SIZE; tail -12 $O/pmc_cfg3_write.md ;;
    trace)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_default -o x -- python $R/bench.py --no-cpu-baseline --no-parity --steps 4 --warmup 1 > $O/trace_default.log 2>&1; echo "trace default rc=$?"
        timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_headline -o x -- python $R/bench.py $Q --steps 20 --warmup 5 > $O/trace_headline.log 2>&1; echo "trace headline rc=$?" )
      for t in default headline; do
        db=$(find $O/trace_$t -name "*results.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/trace_$t.md 2>> $O/trace_err.log
      done
      rm -rf $O/trace_default/ $O/trace_headline/
      grep -o '"kernel_ms_per_step":[0-9.]*' $O/trace_headline.log | head -2; head -12 $O/trace_headline.md ;;
    power)
      python bench.py --workload cfg3 $Q --steps 400 --warmup 2 > $O/power_bench.json 2> $O/power_bench.err &
      BP=$!
      while kill -0 $BP 2>/dev/null; do
        rocm-smi --showpower --showclocks 2>/dev/null | grep -E "GPU\[0\].*(Power|sclk)" | tr '\n' ' ' >> $O/smi.txt; echo >> $O/smi.txt; sleep 0.4
      done
      rocm-smi --showmaxpower 2>/dev/null | grep -E "GPU\[0\]" >> $O/smi.txt; tail -3 $O/smi.txt ;;
    ceiling) timeout 600 python tools/ceiling.py --seconds 8 > $O/ceiling.md 2> $O/ceiling.err; echo "ceiling rc=$?"; cat $O/ceiling.md ;;
    asan)
      ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
      E="env TAVB_LIBRARY=libtavb_debug.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1"
      python tools/asan_exercise.py > $O/plain.txt 2>&1; echo "plain rc=$?"; tail -n 4 $O/plain.txt
      timeout -k 5 600 $E python tools/asan_exercise.py > $O/asan.txt 2>&1; echo "asan rc=$?"
      if ! grep -q "asan exercise: all good" $O/asan.txt; then  # (RCCL under the ASan runtime may refuse to initialise: the rest must still be clean)
        timeout -k 5 600 $E TAVB_ASAN_SKIP_RCCL=1 python tools/asan_exercise.py > $O/asan_norccl.txt 2>&1; echo "asan (no rccl) rc=$?"; tail -n 6 $O/asan_norccl.txt
      fi
      tail -n 12 $O/asan.txt; grep -c "runtime error\|AddressSanitizer" $O/asan.txt ;;
    *) echo "unknown step $step" ;;
  esac
  echo "[$step: $(( $(date +%s) - S )) s]"
done
