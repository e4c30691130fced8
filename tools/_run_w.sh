export TMPDIR=/tmp
ROOT=$PWD
python tools/bench_c1.py 5 200 2>/dev/null | tail -1
cd /tmp; rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o p --output-format csv -- python $ROOT/tools/bench_c1.py 5 200 > /tmp/pp.log 2>&1
python $ROOT/tools/summarize_prof.py /tmp/pp 2>/dev/null | head -16 | cut -c1-150
