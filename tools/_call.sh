export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
cd /tmp
for C in 1 2; do
  rocprofv3 --kernel-trace -d $OUT/r04b_trace_c$C -o t --output-format csv -- python $OLDPWD/tools/bench_decoder_chains.py 256 64 2 -- "chains=$C" > $OUT/r04b_trace_c$C.log 2>&1
  python $OLDPWD/tools/trace_overlap.py $OUT/r04b_trace_c$C 0.6 0.98 > $OUT/r04b_overlap_c$C.txt 2>&1
done
cd $OLDPWD
python tools/bench_decoder_chains.py 512 64 2 -- "chains=1" "chains=2" "chains=2 SMI_DEC_KS_OUT=2 SMI_DEC_LOGITS_GRID=128" > $OUT/r04b_chains_n512.log 2>&1
python -m pytest tests/test_gpu_decoder.py -m gpu -q -x -k chains 2>&1 | tail -5 > $OUT/r04b_pytest_chains.log
find $OUT -name "*kernel_trace.csv" -size +30M -delete
cat $OUT/r04b_trace_c1.log | tail -3; cat $OUT/r04b_overlap_c1.txt; cat $OUT/r04b_trace_c2.log | tail -3; cat $OUT/r04b_overlap_c2.txt; cat $OUT/r04b_chains_n512.log; cat $OUT/r04b_pytest_chains.log
