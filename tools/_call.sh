export TMPDIR=/tmp
mkdir -p gpurun_out
V=$PWD/gpurun_variants
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
bash tools/gpu_exp.sh r03s_dec python tools/bench_decoder.py 256 64 -- "SMI_LIB=$V/prev.so" "SMI_X=1" "SMI_LIB=$V/prev.so" "SMI_X=1"
bash tools/gpu_exp.sh r03s_speech python tools/bench_speech.py -- "SMI_LIB=$V/prev.so" "SMI_X=1"
bash tools/gpu_exp.sh r03s_c1 python tools/bench_c1.py -- "SMI_LIB=$V/prev.so" "SMI_X=1"
bash tools/gpu_exp.sh r03s_enc python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --xsim-n 262144 -- "SMI_LIB=$V/prev.so" "SMI_X=1"
