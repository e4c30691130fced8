mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "xsim or mining or margin" 2>&1 | tail -3
V=$PWD/sonar_amd/lib/variant_head.so
bash tools/gpu_exp.sh r03y1 timeout 300 python tools/probe_xsim.py 262144 1048576 1 -- "SMI_LIB=$V" "SMI_XSIM_TM=1" "SMI_LIB=$V" "SMI_XSIM_TM=1" "SMI_XSIM_TM=0"
bash tools/gpu_exp.sh r03y4 timeout 300 python tools/probe_xsim.py 262144 1048576 4 -- "SMI_LIB=$V" "SMI_XSIM_TM=1"
bash tools/gpu_exp.sh r03y2 timeout 300 python tools/probe_xsim.py 262144 1048576 2 -- "SMI_LIB=$V" "SMI_XSIM_TM=1"
bash tools/gpu_exp.sh r03yf timeout 300 python tools/probe_xsim.py 1048576 1048576 1 -- "SMI_LIB=$V" "SMI_XSIM_TM=1"
