bash tools/gpu_round.sh r03zz pmc > gpurun_out/r03zz_round.log 2>&1
bash tools/gpu_prof_legs.sh r03zz > gpurun_out/r03zz_legs.log 2>&1
python tools/bench_c1.py > gpurun_out/r03zz_c1.log 2>&1
python tools/bench_c1.py 5 100 >> gpurun_out/r03zz_c1.log 2>&1
python tools/bench_e2e.py > gpurun_out/r03zz_e2e.log 2>&1
tail -4 gpurun_out/r03zz_pytest_gpu.log; tail -2 gpurun_out/r03zz_smoke.log; head -8 gpurun_out/r03zz_kernel_stats.txt; cat gpurun_out/r03zz_mfma_util.txt | head -6; tail -3 gpurun_out/r03zz_c1.log; tail -2 gpurun_out/r03zz_e2e.log
