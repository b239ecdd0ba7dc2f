#!/bin/bash
# end to end with the reader's blocks taken wholesale; CLI parity tests
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2o; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "cli or plumbing or scale or binary" 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
bash tools/gpu_r2_e2e.sh 5000000 noref 2>&1 | grep -E "amd_|written"
cd /tmp/e2e && ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> $O/host_timing.err; grep "host " $O/host_timing.err | tail -12
