#!/bin/bash
# first GPU contact: kernel + encoder parity, then a quick timing probe
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
python tools/probe_perf.py > gpurun_out/probe_perf.log 2>&1; tail -20 gpurun_out/probe_perf.log
