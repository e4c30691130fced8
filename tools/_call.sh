bash tools/gpu_round.sh r03g pmc > gpurun_out/r03g_round.log 2>&1
bash tools/gpu_prof_legs.sh r03g > gpurun_out/r03g_legs.log 2>&1
python tools/bench_c1.py > gpurun_out/r03g_c1.log 2>&1
tail -5 gpurun_out/r03g_pytest_gpu.log; tail -3 gpurun_out/r03g_smoke.log; cat gpurun_out/r03g_bench.json; head -20 gpurun_out/r03g_kernel_stats.txt; cat gpurun_out/r03g_pmc_summary.txt | head -30; cat gpurun_out/r03g_mfma_util.txt | head -20; tail -30 gpurun_out/r03g_legs.log; cat gpurun_out/r03g_c1.log | tail -2
