#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp13.log
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py tests/test_gpu_fullsize.py::test_encoder_full_width_vs_oracle tests/test_gpu_fullsize.py::test_baseline_config_properties tests/test_gpu_twin.py -x -q 2>&1 | tail -4 >> $OUT/exp13.log
for v in 0 1 0 1; do
  echo "== SMI_ATT_ORDER=$v" >> $OUT/exp13.log
  SMI_ATT_ORDER=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()}, 'varlen', d['varlen']['ms_per_step'])" >> $OUT/exp13.log
done
cat $OUT/exp13.log
