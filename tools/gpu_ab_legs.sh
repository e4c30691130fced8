#!/bin/bash
# Same-box alternating A/B of tuning switches on the C5 decoder probe, the C4 speech probe and the small-batch probe.
# usage: bash tools/gpu_ab_legs.sh <tag> "<ENV=.. for A>" "<ENV=.. for B>" [reps]
TAG=$1; A=$2; B=$3; REPS=${4:-2}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_legs.txt
: > $LOG
for rep in $(seq $REPS); do
  for cfg in "$A" "$B"; do
    echo "== [$cfg] rep $rep" >> $LOG
    env $cfg python tools/bench_decoder.py 256 64 2>/dev/null | tail -1 >> $LOG
    env $cfg python tools/bench_speech.py 2>/dev/null | tail -1 >> $LOG
    env $cfg python tools/bench_c1.py 2>/dev/null | tail -3 >> $LOG
  done
done
cat $LOG
