#!/bin/bash
# Same-box alternating A/B of the in-tree library against variant builds (gpurun_variants/lib<name>.so) on the C2 step, the C5 decoder
# probe and the C4 speech probe.  usage: bash tools/gpu_ab.sh <tag> <variant name> [pytest -k expression]
TAG=$1; VAR=$PWD/gpurun_variants/lib$2.so; KEXPR=$3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_ab.txt
: > $LOG
if [ -n "$KEXPR" ]; then
  python -m pytest tests -m gpu -q -x -k "$KEXPR" 2>&1 | tail -15 >> $LOG
fi
c2() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('C2 %.2f ms  %.0f sent/s  ' % (d['ms_per_step'], d['value']) + ' '.join('%s %.2f' % (k, v['ms_per_step']) for k, v in d['kernels'].items() if k.startswith('gemm') or k == 'attention'))"; }
for rep in 1 2; do
  for cfg in variant tree; do
    if [ $cfg == variant ]; then export SMI_LIB=$VAR; else unset SMI_LIB; fi
    echo "== $cfg ($2 / in-tree) rep $rep" >> $LOG
    c2 >> $LOG 2>&1
    python tools/bench_decoder.py 256 64 2>/dev/null | tail -1 >> $LOG
    python tools/bench_speech.py 2>/dev/null | tail -1 >> $LOG
  done
done
unset SMI_LIB
cat $LOG
