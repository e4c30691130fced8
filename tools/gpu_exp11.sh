#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp11.log
for r in 0 1 2 3 0; do
  echo "== SMI_REVERSE=$r" >> $OUT/exp11.log
  SMI_REVERSE=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $OUT/exp11.log
done
SMI_REVERSE=3 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py::test_baseline_config_properties tests/test_gpu_fullsize.py::test_encoder_full_depth_vs_oracle tests/test_gpu_encoder.py -x -q 2>&1 | tail -2 >> $OUT/exp11.log
cat $OUT/exp11.log
