#!/bin/bash
# round-2 experiment 2: decoder attention variants (0 plain, 1 XCD-local block order, 2 beams share K/V through LDS)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_sampling.py "tests/test_gpu_fullsize.py::test_basic_decoder_tokens_vs_oracle_full_size" "tests/test_gpu_fullsize.py::test_basic_decoder_greedy_is_teacher_forced_argmax" tests/test_gpu_twin.py -x -q 2>&1 | tail -8 > $OUT/exp2_pytest.log
rm -f $OUT/exp2_decoder.log
for mode in 0 1 2; do
  echo "== SMI_DEC_ATTN=$mode" >> $OUT/exp2_decoder.log
  SMI_DEC_ATTN=$mode python tools/bench_decoder.py 256 64 2>&1 | tail -1 >> $OUT/exp2_decoder.log
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/exp2_dprof -o dec --output-format csv -- python $OLDPWD/tools/bench_decoder.py 256 64 > $OUT/exp2_dprof.log 2>&1
cd $OLDPWD
python tools/summarize_prof.py $OUT/exp2_dprof > $OUT/exp2_decoder_kernel_stats.txt 2>&1
find $OUT/exp2_dprof -name "*kernel_trace*" -delete 2>/dev/null
cat $OUT/exp2_pytest.log | tail -4; cat $OUT/exp2_decoder.log; head -12 $OUT/exp2_decoder_kernel_stats.txt | cut -c1-150
