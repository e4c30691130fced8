export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_speech.py -m gpu -q -x 2>&1 | tail -4 > $OUT/r04j_pytest.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "speech" 2>&1 | tail -4 >> $OUT/r04j_pytest.log
: > $OUT/r04j_relpos.log
for i in 1 2; do
  SMI_LIB=$PWD/gpurun_variants/librelpos_r04g.so python tools/bench_speech.py >> $OUT/r04j_relpos.log 2>&1
  python tools/bench_speech.py >> $OUT/r04j_relpos.log 2>&1
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/r04j_prof_speech -o s --output-format csv -- python $OLDPWD/tools/bench_speech.py > /dev/null 2>&1
cd $OLDPWD
python tools/summarize_prof.py $OUT/r04j_prof_speech > $OUT/r04j_speech_kernel_stats.txt 2>&1
find $OUT/r04j_prof_speech -name "*kernel_trace*" -delete 2>/dev/null
cat $OUT/r04j_pytest.log; grep "speech n" $OUT/r04j_relpos.log; grep "relpos" $OUT/r04j_speech_kernel_stats.txt | cut -c1-150
