mkdir -p gpurun_out
V=$PWD/sonar_amd/lib/variant_head.so
bash tools/gpu_exp.sh r03h1d timeout 300 python tools/bench_decoder.py 256 64 -- "SMI_LIB=$V" "SMI_X=1" "SMI_LIB=$V" "SMI_X=1"
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/r03h1_prof -o h1 --output-format csv -- python $OLDPWD/tools/bench_decoder.py 256 64 > /dev/null 2>&1; cd $OLDPWD
python tools/summarize_prof.py gpurun_out/r03h1_prof > gpurun_out/r03h1_decoder_kernel_stats.txt 2>&1; find gpurun_out/r03h1_prof -name "*kernel_trace*" -delete; grep "vocab_select\|beam_step" gpurun_out/r03h1_decoder_kernel_stats.txt | cut -c1-150
