export TMPDIR=/tmp
mkdir -p gpurun_out
SMI_LIB=$PWD/gpurun_variants/trace.so python tools/gemm_trace.py 2>&1 | tail -5
