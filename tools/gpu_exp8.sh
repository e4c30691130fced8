#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp8.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> $OUT/exp8.log
for r in 0 2; do
  echo "== SMI_G2_RASTER=$r" >> $OUT/exp8.log
  SMI_G2_RASTER=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-xsim 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()}, 'varlen', d['varlen']['ms_per_step'], 'speech', d['speech']['ms'], 'decoder', d['decoder']['ms_per_step'])" >> $OUT/exp8.log
done
cat $OUT/exp8.log
