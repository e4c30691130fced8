export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py -m gpu -q -x 2>&1 | tail -4 > $OUT/r04n_pytest.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "encoder or speech" 2>&1 | tail -4 >> $OUT/r04n_pytest.log
: > $OUT/r04n_oldv.log
for i in 1 2; do
  for V in old new; do
    if [ $V == old ]; then export SMI_LIB=$PWD/gpurun_variants/libg2old.so; else unset SMI_LIB; fi
    echo "== $V" >> $OUT/r04n_oldv.log
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('C2 ms/step', round(d['ms_per_step'],2), 'sent/s', round(d['value']), {n:round(v.get('ms_per_step', v.get('ms',0)),3) for n,v in k.items() if isinstance(v,dict)})
" >> $OUT/r04n_oldv.log 2>&1
    python tools/bench_speech.py 2>/dev/null | grep "speech n" >> $OUT/r04n_oldv.log
  done
done
unset SMI_LIB
cat $OUT/r04n_pytest.log $OUT/r04n_oldv.log
