#!/bin/bash
mkdir -p gpurun_out/r03_r
timeout 1200 python -m pytest tests/test_gpu_tex.py -x -q > gpurun_out/r03_r/pytest.log 2>&1; tail -3 gpurun_out/r03_r/pytest.log
