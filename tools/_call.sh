export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16; do
  SMI_G2_DUO=3 timeout 300 rocprofv3 --pmc $C GRBM_GUI_ACTIVE --kernel-trace -d $OUT/d3_pmc_$C -o d3 --output-format csv -- python $R/tools/probe_duo.py > /dev/null 2> $OUT/d3_pmc_$C.err
done
SMI_G2_DUO=3 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/d3_sq -o d3 --output-format csv -- python $R/tools/probe_duo.py > /dev/null 2> $OUT/d3_sq.err
cd $R
python tools/summarize_mfma.py $OUT/d3_pmc_SQ_VALU_MFMA_BUSY_CYCLES $OUT/d3_pmc_SQ_INSTS_VALU_MFMA_MOPS_F16 > $OUT/r03d3_mfma.txt 2>&1
python tools/summarize_sq.py $OUT/d3_sq > $OUT/r03d3_sq.txt 2>&1
rm -rf $OUT/d3_pmc_* $OUT/d3_sq
cat $OUT/r03d3_mfma.txt | cut -c1-200; cat $OUT/r03d3_sq.txt | cut -c1-400; tail -3 $OUT/d3_sq.err
