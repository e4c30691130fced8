#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp7.log
for r in 0 1 2; do
  echo "== SMI_G2_RASTER=$r" >> $OUT/exp7.log
  SMI_G2_RASTER=$r python tools/probe_perf.py gemm 2>&1 | grep "tm M=131072" >> $OUT/exp7.log
  SMI_G2_RASTER=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $OUT/exp7.log
done
for st in 0 1; do
  SMI_XSIM_STAGGER=$st python tools/probe_xsim.py 262144 1048576 1 2>&1 | tail -1 | sed "s/^/STAGGER=$st /" >> $OUT/exp7.log
done
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_encoder.py tests/test_gpu_xsim_margin.py tests/test_gpu_speech.py -x -q 2>&1 | tail -2 >> $OUT/exp7.log
cat $OUT/exp7.log
