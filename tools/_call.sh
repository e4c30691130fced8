export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fullsize.py tests/test_gpu_decoder.py tests/test_gpu_twin.py -m gpu -q -x -k "not speech" 2>&1 | tail -4
bash tools/gpu_exp.sh r03t_c1 python tools/bench_c1.py -- "SMI_ENC_SB=0" "SMI_ENC_SB=1" "SMI_ENC_SB=0" "SMI_ENC_SB=1"
bash tools/gpu_exp.sh r03t_b5 python tools/bench_c1.py 5 100 -- "SMI_ENC_SB=0" "SMI_ENC_SB=1"
bash tools/gpu_exp.sh r03t_b128 python tools/bench_c1.py 128 30 -- "SMI_ENC_SB=0" "SMI_ENC_SB=1"
