#!/bin/bash
# -g on the device (gpu tests), then rocprofv3 kernel stats of the partitioned code path on one rank.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dist
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 70 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -k graphviz > $O/pytest_gpu_graph.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_graph.log; tail -6 $O/pytest_gpu_graph.log
cd /tmp
ABG_FORCE_DIST=1 timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o trace -- python $R/bench.py --no-cpu-baseline --warmup 0 > /tmp/prof_trace.log 2>&1
find /tmp/prof_trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_forced_partitioned.csv \;
head -16 $O/kernel_stats_forced_partitioned.csv | cut -c1-170
tail -2 /tmp/prof_trace.log | cut -c1-300
