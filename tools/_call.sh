export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_lowdim.py tests/test_gpu_encoder.py -m gpu -q -x -s 2>&1 | tail -40 > gpurun_out/r03f_pytest.log
cat gpurun_out/r03f_pytest.log
