#!/bin/bash
# Same-box alternating A/B of variant libraries on the C5 decoder probe.  usage: bash tools/gpu_ab_dec.sh <tag> "<variant names...>" [pytest -k expr]
TAG=$1; VARS=$2; KEXPR=$3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_ab.txt
: > $LOG
if [ -n "$KEXPR" ]; then
  python -m pytest tests -m gpu -q -x -k "$KEXPR" 2>&1 | tail -15 >> $LOG
fi
for rep in 1 2; do
  for v in $VARS tree; do
    if [ $v == tree ]; then unset SMI_LIB; else export SMI_LIB=$PWD/gpurun_variants/lib$v.so; fi
    echo "== $v rep $rep" >> $LOG
    python tools/bench_decoder.py 256 64 2>/dev/null | tail -1 >> $LOG
  done
done
unset SMI_LIB
cat $LOG
