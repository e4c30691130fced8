mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "xsim or mining or margin or fullsize" 2>&1 | tail -3
V=$PWD/sonar_amd/lib/variant_head.so
for k in 1 2 4; do bash tools/gpu_exp.sh r03l$k timeout 300 python tools/probe_xsim.py 262144 1048576 $k -- "SMI_LIB=$V" "SMI_X=1" "SMI_LIB=$V" "SMI_X=1" | grep -v amdgpu; done
