#!/bin/bash
# contigs delivered beside the next batch's device work: gpu tests of the CLI / parity, then end-to-end phases
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2x; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
bash tools/gpu_r2_p.sh 2>&1 | tail -6
