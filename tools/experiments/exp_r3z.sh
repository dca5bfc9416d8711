#!/bin/bash
mkdir -p gpurun_out/r03_zz
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
timeout 900 python -m pytest tests/test_gpu_stress.py -x -q > gpurun_out/r03_zz/pytest.log 2>&1; tail -25 gpurun_out/r03_zz/pytest.log
