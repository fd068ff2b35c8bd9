#!/bin/bash
# round 4: the N > 1 code path of bench.py on ONE rank (own RCCL communicator, forced collective), and the driver's torchrun form with one rank
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4/dist1; mkdir -p $O
TAVB_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload cfg3 --rows 1250000 --steps 40 --no-cpu-baseline > $O/one_rank_forced.json 2> $O/one_rank_forced.err; echo "forced rc=$?"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-sub --no-cpu-baseline > $O/torchrun_one_rank.json 2> $O/torchrun_one_rank.err; echo "torchrun rc=$?"
python - <<'PY'
import json
for n in ("one_rank_forced","torchrun_one_rank"):
    try:
        d=json.loads(open(f"gpurun_out/r4/dist1/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["n_gpus"], d["config"]["parallelism"], d["roofline"]["frac"], d["parity"]["ok"], d["roofline"].get("other_kernels_ms_per_step"))
    except Exception as e: print(n, "ERR", e)
PY
tail -3 $O/one_rank_forced.err; tail -3 $O/torchrun_one_rank.err
