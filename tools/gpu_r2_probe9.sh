#!/bin/bash
mkdir -p gpurun_out/r2p9
python -m pytest tests/test_gpu_parity.py tests/test_multidevice.py -x -q -m gpu -k "survivors or predicate or paging or device_group or subset" > gpurun_out/r2p9/pytest.txt 2>&1
tail -n 15 gpurun_out/r2p9/pytest.txt
