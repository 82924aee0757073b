#!/bin/bash
mkdir -p gpurun_out/r06
export PYTHONPATH=$PWD
timeout 1700 python -m pytest tests/test_tracking_gpu.py tests/test_fusion_gpu.py tests/test_ref_live_gpu.py tests/test_ref_pin_gpu.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06/pytest_4.txt
cat gpurun_out/r06/pytest_4.txt
scripts/ab_env.sh 2 "X=0" "DMS_CLEAN_INLINE_SCAN_MAX=4096"
