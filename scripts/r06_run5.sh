#!/bin/bash
mkdir -p gpurun_out/r06
export PYTHONPATH=$PWD
timeout 1700 python -m pytest tests/test_fusion_gpu.py tests/test_tracking_gpu.py tests/test_edge_cases_gpu.py tests/test_clusters_gpu.py tests/test_ref_cf_pin_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06/pytest_5.txt
cat gpurun_out/r06/pytest_5.txt
scripts/ab_env.sh 2 "DMS_SO3_BESIDE_MODEL=0" "DMS_SO3_BESIDE_MODEL=1"
scripts/r06_phases.sh "DMS_SO3_BESIDE_MODEL=0" "DMS_SO3_BESIDE_MODEL=1"
