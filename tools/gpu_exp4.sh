#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fullsize.py::test_baseline_config_properties tests/test_gpu_fullsize.py::test_encoder_full_depth_vs_oracle tests/test_gpu_twin.py -x -q 2>&1 | tail -3 > $OUT/exp4.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $OUT/exp4.log
cat $OUT/exp4.log
