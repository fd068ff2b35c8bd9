#!/bin/bash
# round 2, probe 7: whole GPU suite after multi-device / re-rank / ABI additions
mkdir -p gpurun_out/r2p7
python -m pytest tests/test_multidevice.py tests/test_sharded_gloo.py -x -q -m gpu > gpurun_out/r2p7/pytest.txt 2>&1
tail -n 30 gpurun_out/r2p7/pytest.txt
