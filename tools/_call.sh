export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r03d_pytest.log
cat gpurun_out/r03d_pytest.log
V=$PWD/gpurun_variants
bash tools/gpu_exp.sh r03d_dec python tools/bench_decoder.py 256 64 -- "SMI_LIB=$V/shfl.so SMI_DEC_KS_OUT=4" "SMI_DEC_KS_OUT=4" "SMI_DEC_KS_OUT=2" "SMI_DEC_KS_OUT=1" "SMI_LIB=$V/shfl.so SMI_DEC_KS_OUT=2"
bash tools/gpu_exp.sh r03d_speech python tools/bench_speech.py -- "SMI_LIB=$V/shfl.so" "SMI_X=1"
bash tools/gpu_exp.sh r03d_c1 python tools/bench_c1.py -- "SMI_LIB=$V/shfl.so" "SMI_X=1"
bash tools/gpu_exp.sh r03d_enc python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-xsim --no-extras -- "SMI_LIB=$V/shfl.so" "SMI_X=1"
