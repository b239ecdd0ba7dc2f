#!/bin/bash
# last check of the round: every gpu test, smoke, the default bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2s_final; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json; echo
