#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp12.log
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py tests/test_gpu_fullsize.py::test_encoder_full_width_vs_oracle tests/test_gpu_fullsize.py::test_encoder_full_depth_vs_oracle tests/test_gpu_fullsize.py::test_baseline_config_properties tests/test_gpu_twin.py tests/test_gpu_checkpoint_ingest.py -x -q 2>&1 | tail -6 >> $OUT/exp12.log
for v in 0 1 0 1; do
  echo "== SMI_ENC_X_TM=$v" >> $OUT/exp12.log
  SMI_ENC_X_TM=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $OUT/exp12.log
done
cat $OUT/exp12.log
