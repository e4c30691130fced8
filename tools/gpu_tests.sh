#!/bin/bash
# GPU pass without profiling: parity tests (verbose failures), smoke, bench.  usage: bash tools/gpu_tests.sh <tag> [pytest args]
TAG=${1:-r03}
shift
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -q -s "$@" > $OUT/${TAG}_pytest_gpu_full.log 2>&1
tail -60 $OUT/${TAG}_pytest_gpu_full.log > $OUT/${TAG}_pytest_gpu.log
grep -E "max \(1 - cos|token-identical|near-ties|logits: max|engine errors|cosine matrix|dot products|basic decoder|english speech|max \|logit diff|decisions equal|forced steps|frames" $OUT/${TAG}_pytest_gpu_full.log > $OUT/${TAG}_pytest_measured.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1
python bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -25 $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_pytest_measured.log | tail -60; tail -3 $OUT/${TAG}_smoke.log; cat $OUT/${TAG}_bench.json; tail -5 $OUT/${TAG}_bench.err
