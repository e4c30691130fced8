#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_speech.py tests/test_gpu_fullsize.py::test_speech_encoder_english_vs_oracle_full_size tests/test_gpu_fullsize.py::test_speech_encoder_full_size_properties tests/test_gpu_twin.py tests/test_gpu_heads.py tests/test_gpu_checkpoint_ingest.py -x -q 2>&1 | tail -3 > $OUT/exp10.log
python tools/bench_speech.py 2>&1 | tail -3 >> $OUT/exp10.log
cat $OUT/exp10.log
