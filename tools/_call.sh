export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_encoder.py -m gpu -q -x -s -k "not decoder and not speech" 2>&1 | grep -E "passed|failed|1 - cos" | tail -12
bash tools/gpu_exp.sh r03n_enc python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-xsim --no-extras -- "SMI_ENC_LNFOLD=0" "SMI_ENC_LNFOLD=1" "SMI_ENC_LNFOLD=2" "SMI_ENC_LNFOLD=0" "SMI_ENC_LNFOLD=2"
