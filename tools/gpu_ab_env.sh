#!/bin/bash
# Same-box alternating A/B of tuning switches on the C2 step.  usage: bash tools/gpu_ab_env.sh <tag> "<ENV=.. for A>" "<ENV=.. for B>" [reps]
TAG=$1; A=$2; B=$3; REPS=${4:-2}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_ab.txt
: > $LOG
c2() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-xsim 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('C2 %.2f ms  %.0f sent/s  ' % (d['ms_per_step'], d['value']) + ' '.join('%s %.2f' % (k, v['ms_per_step']) for k, v in d['kernels'].items() if k.startswith('gemm') or k == 'attention'))"; }
for rep in $(seq $REPS); do
  for cfg in "$A" "$B"; do
    echo "== [$cfg] rep $rep" >> $LOG
    env $cfg bash -c "$(declare -f c2); c2" >> $LOG 2>&1
  done
done
cat $LOG
