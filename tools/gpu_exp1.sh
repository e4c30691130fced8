#!/bin/bash
# round-2 experiment 1: decoder (beam-grouped attention, sparse logits store) + xsim raster knobs
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_sampling.py tests/test_gpu_xsim_margin.py "tests/test_gpu_fullsize.py::test_basic_decoder_tokens_vs_oracle_full_size" "tests/test_gpu_fullsize.py::test_basic_decoder_greedy_is_teacher_forced_argmax" -x -q -s 2>&1 | tail -25 > $OUT/exp1_pytest.log
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "== ATTN_GROUP=$1 SPARSE_LOGITS=$2" >> $OUT/exp1_decoder.log
  SMI_DEC_ATTN_GROUP=$1 SMI_DEC_SPARSE_LOGITS=$2 python tools/bench_decoder.py 256 64 2>&1 | tail -2 >> $OUT/exp1_decoder.log
done
for gm in 2 4 8 16; do for ch in 8 16; do
  SMI_XSIM_GM=$gm SMI_XSIM_CHUNKS=$ch python tools/probe_xsim.py 262144 1048576 1 2>&1 | tail -1 >> $OUT/exp1_xsim.log
done; done
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/exp1_dprof -o dec --output-format csv -- python $OLDPWD/tools/bench_decoder.py 256 64 > $OUT/exp1_dprof.log 2>&1
cd $OLDPWD
python tools/summarize_prof.py $OUT/exp1_dprof > $OUT/exp1_decoder_kernel_stats.txt 2>&1
find $OUT/exp1_dprof -name "*kernel_trace*" -delete 2>/dev/null
cat $OUT/exp1_pytest.log | tail -8; cat $OUT/exp1_decoder.log; cat $OUT/exp1_xsim.log; head -16 $OUT/exp1_decoder_kernel_stats.txt | cut -c1-140
