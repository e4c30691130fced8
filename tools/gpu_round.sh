#!/bin/bash
# Standard GPU pass: parity tests, smoke, bench, rocprofv3 kernel stats (+ optional PMC passes).
# usage: bash tools/gpu_round.sh [tag] [pmc]
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1
python bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o ${TAG} --output-format csv -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
cd $OLDPWD
python tools/summarize_prof.py $OUT/${TAG}_prof > $OUT/${TAG}_kernel_stats.txt 2>&1
if [ "$2" == "pmc" ]; then
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace -d $OUT/${TAG}_pmc_$C -o ${TAG} --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-xsim --no-extras > /dev/null 2> $OUT/${TAG}_pmc_$C.err
  done
  cd $OLDPWD
  python tools/summarize_pmc.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE > $OUT/${TAG}_pmc_summary.txt 2>&1
  # raw per-dispatch CSVs are large; keep only the summaries
  rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
fi
if [ "$2" == "pmc" ]; then
  # MFMA utilisation / hardware-counted flops / effective clock of the GEMM stack (SQ + GRBM counters only,
  # one pass per SQ counter so that an unknown counter name cannot void the other)
  cd /tmp
  for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16; do
    rocprofv3 --pmc $C GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc_$C -o ${TAG} --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --xsim-n 131072 > /dev/null 2> $OUT/${TAG}_pmc_$C.err
  done
  cd $OLDPWD
  python tools/summarize_mfma.py $OUT/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES $OUT/${TAG}_pmc_SQ_INSTS_VALU_MFMA_MOPS_F16 > $OUT/${TAG}_mfma_util.txt 2>&1
  rm -rf $OUT/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES $OUT/${TAG}_pmc_SQ_INSTS_VALU_MFMA_MOPS_F16
fi
find $OUT/${TAG}_prof -name "*kernel_trace*" -delete 2>/dev/null
tail -5 $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_smoke.log | tail -3; cat $OUT/${TAG}_bench.json; head -30 $OUT/${TAG}_kernel_stats.txt
