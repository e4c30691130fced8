export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
: > $OUT/r04m_hot.log
cd /tmp
for V in hot cur; do
  if [ $V == hot ]; then export SMI_LIB=$OLDPWD/gpurun_variants/librelpos_hot.so; else unset SMI_LIB; fi
  rocprofv3 --kernel-trace --stats -d $OUT/r04m_prof_$V -o s --output-format csv -- python $OLDPWD/tools/bench_speech.py >> $OUT/r04m_hot.log 2>&1
  python $OLDPWD/tools/summarize_prof.py $OUT/r04m_prof_$V 2>&1 | grep "relpos\|dwconv" | cut -c1-150 >> $OUT/r04m_hot.log
  find $OUT/r04m_prof_$V -name "*kernel_trace*" -delete 2>/dev/null
done
cd $OLDPWD
grep "speech n\|relpos\|dwconv" $OUT/r04m_hot.log
