#!/bin/bash
mkdir -p gpurun_out/r2p8
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "load_paths or adopted or rerank or device_only or appends or message" > gpurun_out/r2p8/pytest.txt 2>&1
tail -n 15 gpurun_out/r2p8/pytest.txt
python tools/load_bench.py > gpurun_out/r2p8/load_bench.jsonl 2> gpurun_out/r2p8/load_bench.err
cat gpurun_out/r2p8/load_bench.jsonl; tail -n 3 gpurun_out/r2p8/load_bench.err
