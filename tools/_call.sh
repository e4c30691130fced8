mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "decoder or beam or attention or speech or kernels or sampling or encoder" 2>&1 | tail -3 > gpurun_out/r03f1_tests.log; cat gpurun_out/r03f1_tests.log
V=$PWD/sonar_amd/lib/variant_head.so
bash tools/gpu_exp.sh r03f1d timeout 300 python tools/bench_decoder.py 256 64 -- "SMI_LIB=$V" "SMI_X=1" "SMI_LIB=$V" "SMI_X=1"
bash tools/gpu_exp.sh r03f1s timeout 300 python tools/bench_speech.py -- "SMI_LIB=$V" "SMI_X=1" "SMI_LIB=$V" "SMI_X=1"
bash tools/gpu_exp.sh r03f1e timeout 300 python tools/probe_perf.py -- "SMI_LIB=$V" "SMI_X=1" "SMI_LIB=$V" "SMI_X=1"
