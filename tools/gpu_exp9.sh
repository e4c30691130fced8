#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp9.log
for k in 0 1; do
  echo "== SMI_G2_KSTAGGER=$k" >> $OUT/exp9.log
  SMI_G2_KSTAGGER=$k python tools/probe_perf.py gemm 2>&1 | grep "tm M=131072" >> $OUT/exp9.log
  SMI_G2_KSTAGGER=$k python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $OUT/exp9.log
done
SMI_G2_KSTAGGER=1 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -2 >> $OUT/exp9.log
cat $OUT/exp9.log
