#!/bin/bash
# diagnosis: ABG_MEMO_VERIFY recomputes every memo hit
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
for v in 1 0; do
echo "== presearch=$v verify"
ABG_PRESEARCH=$v ABG_MEMO_VERIFY=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "reproduces" 2>&1 | grep -v "^$" | tail -25 | cut -c1-300
done
echo "== presearch=1 no verify, p2 small"
ABG_PRESEARCH=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "reproduces" 2>&1 | grep -v "^$" | tail -30 | cut -c1-400
