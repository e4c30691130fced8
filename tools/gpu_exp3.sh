#!/bin/bash
# round-2 experiment 3: counted vmcnt at tile start (+ no end-of-tile barrier) in the 256x256 engine
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp3.log
for v in base counted counted_nobar; do
  if [ $v == base ]; then unset SMI_LIB; else export SMI_LIB=$PWD/gpurun_variants/$v.so; fi
  echo "== $v" >> $OUT/exp3.log
  python tools/probe_perf.py gemm 2>&1 | grep "tm M" >> $OUT/exp3.log
  python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py::test_encoder_full_width_vs_oracle tests/test_gpu_fullsize.py::test_baseline_config_properties tests/test_gpu_encoder.py -x -q 2>&1 | tail -2 >> $OUT/exp3.log
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --xsim-n 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'xsim', d['xsim']['pairs_per_s'])" >> $OUT/exp3.log
done
cat $OUT/exp3.log
