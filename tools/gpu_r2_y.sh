#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2y; mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json; echo
bash tools/gpu_r2_e2e.sh 5000000 noref 2>&1 | grep -E "amd_"
