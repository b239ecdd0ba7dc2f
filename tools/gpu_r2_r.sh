#!/bin/bash
# the default bench line with the overlapped PASS 1 accounted once, and configs[2] again
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2r; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json; echo
timeout 1500 python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_config2.json 2> $O/bench_config2.err; cut -c1-300 $O/bench_config2.json; echo; tail -2 $O/bench_config2.err
