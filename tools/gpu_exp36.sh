#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp36.log
for v in base new base new base new; do
  if [ $v = new ]; then unset SMI_LIB; else export SMI_LIB=$PWD/gpurun_variants/lib_base.so; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-xsim --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v bench', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items() if k in ('layernorm','gemm_ffn1','gemm_qkv')})" >> $OUT/exp36.log
done
cat $OUT/exp36.log
