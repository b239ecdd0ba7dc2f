#!/bin/bash
# End-of-round evidence on a gpurun box: gpu tests, smoke, the default bench line, the rocprofv3
# kernel trace of the same command and the HBM traffic counters.  Summaries -> gpurun_out/final/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o trace -- python $R/bench.py --no-cpu-baseline > /tmp/prof_trace.log 2>&1
find /tmp/prof_trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -12 $O/kernel_stats.csv | cut -c1-160
cd $R
bash tools/gpu_pmc_traffic.sh --warmup 0 --steps 1 --no-cpu-baseline > $O/pmc.log 2>&1; cp gpurun_out/prof/pmc_traffic.json $O/ 2>/dev/null; tail -5 $O/pmc.log | cut -c1-200
