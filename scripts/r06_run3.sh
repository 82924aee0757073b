#!/bin/bash
mkdir -p gpurun_out/r06
export PYTHONPATH=$PWD
timeout 1700 python -m pytest tests/test_tracking_gpu.py tests/test_fusion_gpu.py tests/test_edge_cases_gpu.py tests/test_ref_pin_gpu.py tests/test_refframe_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06/pytest_3.txt
cat gpurun_out/r06/pytest_3.txt
scripts/abn.sh 2 "DMS_TRACK_FUSE=0" "DMS_TRACK_FUSE=1"
scripts/ab_env.sh 2 "DMS_TRACK_FUSE=0" "DMS_TRACK_FUSE=1"
