bash tools/gpu_round.sh r03zx pmc > gpurun_out/r03zx_round.log 2>&1
bash tools/gpu_prof_legs.sh r03zx > gpurun_out/r03zx_legs.log 2>&1
tail -4 gpurun_out/r03zx_pytest_gpu.log; tail -2 gpurun_out/r03zx_smoke.log; head -6 gpurun_out/r03zx_kernel_stats.txt; cat gpurun_out/r03zx_mfma_util.txt | head -5; tail -1 gpurun_out/r03zx_decoder.log; tail -1 gpurun_out/r03zx_speech.log
