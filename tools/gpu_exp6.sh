#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp6.log
for r in 0 1 2; do
  echo "== SMI_G2_RASTER=$r" >> $OUT/exp6.log
  SMI_G2_RASTER=$r python tools/probe_perf.py gemm 2>&1 | grep "tm M=131072 N=8192" >> $OUT/exp6.log
  SMI_G2_RASTER=$r python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py::test_encoder_full_width_vs_oracle tests/test_gpu_fullsize.py::test_baseline_config_properties -x -q 2>&1 | tail -1 >> $OUT/exp6.log
  SMI_G2_RASTER=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $OUT/exp6.log
done
cat $OUT/exp6.log
