#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
SMI_LIB=$PWD/gpurun_variants/trace.so python tools/gemm_trace.py > $OUT/exp5.log 2>&1
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fullsize.py::test_baseline_config_properties tests/test_gpu_fullsize.py::test_xsim_large_known_neighbours tests/test_gpu_xsim_margin.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3 >> $OUT/exp5.log
python tools/probe_xsim.py 262144 1048576 1 2>&1 | tail -1 >> $OUT/exp5.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $OUT/exp5.log
cat $OUT/exp5.log
