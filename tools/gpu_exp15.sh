#!/bin/bash
# r02 exp 15b: pair-map row kernels on the tile-major residual stream, final shapes
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp15.log
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_kernels.py tests/test_gpu_twin.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 >> $OUT/exp15.log
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-xsim --no-extras"
for x in 1 0 1 0; do
  SMI_ENC_X_TM=$x $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={n: round(v['ms_per_step'],4) for n,v in d['kernels'].items()}
print('X_TM=$x', d['value'], d['ms_per_step'], {n: k[n] for n in ('embed','layernorm','ln_pool')})" >> $OUT/exp15.log
done
cat $OUT/exp15.log
