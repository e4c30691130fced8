export TMPDIR=/tmp
mkdir -p gpurun_out
V=$PWD/gpurun_variants
timeout 900 python -m pytest tests/test_gpu_speech.py tests/test_gpu_fullsize.py -m gpu -q -x -k "speech" 2>&1 | tail -3
bash tools/gpu_exp.sh r03x_speech python tools/bench_speech.py -- "SMI_LIB=$V/prev.so" "SMI_X=1" "SMI_LIB=$V/prev.so" "SMI_X=1"
