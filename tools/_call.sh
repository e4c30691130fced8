export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_xsim_margin.py tests/test_gpu_decoder.py tests/test_gpu_encoder.py tests/test_gpu_speech.py tests/test_gpu_rccl.py -m gpu -q -x 2>&1 | tail -8 > $OUT/r04c_pytest.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "xsim" 2>&1 | tail -4 >> $OUT/r04c_pytest.log
python -m pytest tests -m gpu -q -x -k "xsim or topk" 2>&1 | tail -4 >> $OUT/r04c_pytest.log
: > $OUT/r04c_xsim_ll.log
for K in 1 2 4 8; do
  for LL in 0 1; do
    SMI_XSIM_LL=$LL python tools/probe_xsim.py 262144 1048576 $K >> $OUT/r04c_xsim_ll.log 2>&1
  done
done
python tools/bench_decoder_chains.py 1024 64 2 -- "chains=1" "chains=2" "chains=3" "chains=4" "chains=0" > $OUT/r04c_chains_n1024.log 2>&1
python tools/bench_decoder_chains.py 768 64 2 -- "chains=1" "chains=2" "chains=3" "chains=0" > $OUT/r04c_chains_n768.log 2>&1
cat $OUT/r04c_pytest.log; grep xsim $OUT/r04c_xsim_ll.log; cat $OUT/r04c_chains_n1024.log $OUT/r04c_chains_n768.log | grep chains
