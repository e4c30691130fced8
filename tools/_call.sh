mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_sampling.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r03e1_tests.log
cat gpurun_out/r03e1_tests.log
timeout 900 python -m pytest tests -m gpu -x -q -k "xsim or mining or margin" 2>&1 | tail -3
