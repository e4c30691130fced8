export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_speech.py -m gpu -q -x 2>&1 | tail -4 > $OUT/r04o_pytest.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "speech" 2>&1 | tail -4 >> $OUT/r04o_pytest.log
: > $OUT/r04o_dw.log
for i in 1 2; do
  SMI_LIB=$PWD/gpurun_variants/libdw_head.so python tools/bench_speech.py 2>/dev/null | grep "speech n" >> $OUT/r04o_dw.log
  python tools/bench_speech.py 2>/dev/null | grep "speech n" >> $OUT/r04o_dw.log
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/r04o_prof -o s --output-format csv -- python $OLDPWD/tools/bench_speech.py > /dev/null 2>&1
cd $OLDPWD
python tools/summarize_prof.py $OUT/r04o_prof 2>&1 | grep "dwconv\|relpos" | cut -c1-150 >> $OUT/r04o_dw.log
find $OUT/r04o_prof -name "*kernel_trace*" -delete 2>/dev/null
cat $OUT/r04o_pytest.log $OUT/r04o_dw.log
