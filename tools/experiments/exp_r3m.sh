#!/bin/bash
# full GPU suite + smoke
mkdir -p gpurun_out/r03_m
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03_m/pytest_gpu.log 2>&1; tail -4 gpurun_out/r03_m/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
