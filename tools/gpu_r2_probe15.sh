#!/bin/bash
mkdir -p gpurun_out/r2p15
python -m pytest tests -x -q -m gpu > gpurun_out/r2p15/pytest.txt 2>&1
tail -n 8 gpurun_out/r2p15/pytest.txt
