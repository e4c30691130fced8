mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "decoder or beam or generat or lowdim or twin" 2>&1 | tail -3 > gpurun_out/r03g1_tests.log; cat gpurun_out/r03g1_tests.log
bash tools/gpu_exp.sh r03g1d timeout 300 python tools/bench_decoder.py 256 64 -- "SMI_DEC_STATS_TR=0" "SMI_DEC_STATS_TR=1" "SMI_DEC_STATS_TR=0" "SMI_DEC_STATS_TR=1"
