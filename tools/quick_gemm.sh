#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm" 2>&1 | tail -4
for v in ${VARS:-0}; do echo "== VAR $v"; SMI_GEMM_VAR=$v python tools/probe_perf.py gemm 2>&1 | grep gemm; done | tee gpurun_out/probe_gemm_var.log
