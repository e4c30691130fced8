export TMPDIR=/tmp
mkdir -p gpurun_out
V=$PWD/gpurun_variants
bash tools/gpu_exp.sh r03v_enc python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-xsim --no-extras -- "SMI_ENC_LNFOLD=0" "SMI_ENC_LNFOLD=2" "SMI_LIB=$V/fold_plain.so" "SMI_ENC_LNFOLD=2" "SMI_LIB=$V/fold_plain.so"
