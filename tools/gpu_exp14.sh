#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/exp14.log
python -m pytest tests/test_gpu_speech.py tests/test_gpu_fullsize.py::test_speech_encoder_english_vs_oracle_full_size tests/test_gpu_fullsize.py::test_speech_encoder_full_size_properties tests/test_gpu_twin.py -x -q 2>&1 | tail -3 >> $OUT/exp14.log
for v in 0 1 0 1; do
  SMI_RELPOS_ORDER=$v python tools/bench_speech.py 2>&1 | tail -1 | sed "s/^/ORDER=$v /" >> $OUT/exp14.log
done
cat $OUT/exp14.log
