export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_speech.py -m gpu -q -x 2>&1 | tail -6 > $OUT/r04e_pytest.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "speech" 2>&1 | tail -4 >> $OUT/r04e_pytest.log
: > $OUT/r04e_midtm.log
for i in 1 2; do
  SMI_SPEECH_MID_TM=0 python tools/bench_speech.py >> $OUT/r04e_midtm.log 2>&1
  python tools/bench_speech.py >> $OUT/r04e_midtm.log 2>&1
done
cat $OUT/r04e_pytest.log; grep "speech n" $OUT/r04e_midtm.log
