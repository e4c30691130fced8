#!/bin/bash
# Same-box A/B of environment switches on one probe.  usage: bash tools/gpu_exp.sh <tag> <probe cmd...> -- "<ENV=.. ENV=..>" "<...>" ...
# Every configuration runs the probe once; stdout tails go to gpurun_out/<tag>_exp.txt
TAG=$1; shift
PROBE=()
while [ "$1" != "--" ]; do PROBE+=("$1"); shift; done
shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
for CFG in "$@"; do
  echo "== $CFG" >> $OUT/${TAG}_exp.txt
  env $CFG "${PROBE[@]}" 2>&1 | tail -3 >> $OUT/${TAG}_exp.txt
done
cat $OUT/${TAG}_exp.txt
