#!/bin/bash
# rocprofv3 kernel stats of the C4 (speech) and C5 (decoder) probes.  usage: bash tools/gpu_prof_legs.sh [tag]
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_speech -o ${TAG} --output-format csv -- python $ROOT/tools/bench_speech.py > $OUT/${TAG}_speech.log 2> $OUT/${TAG}_speech.err
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_decoder -o ${TAG} --output-format csv -- python $ROOT/tools/bench_decoder.py 256 64 > $OUT/${TAG}_decoder.log 2> $OUT/${TAG}_decoder.err
cd $ROOT
python tools/summarize_prof.py $OUT/${TAG}_prof_speech > $OUT/${TAG}_speech_kernel_stats.txt 2>&1
python tools/summarize_prof.py $OUT/${TAG}_prof_decoder > $OUT/${TAG}_decoder_kernel_stats.txt 2>&1
find $OUT/${TAG}_prof_speech $OUT/${TAG}_prof_decoder -name "*kernel_trace*" -delete 2>/dev/null
tail -2 $OUT/${TAG}_speech.log; head -24 $OUT/${TAG}_speech_kernel_stats.txt; tail -2 $OUT/${TAG}_decoder.log; head -16 $OUT/${TAG}_decoder_kernel_stats.txt
