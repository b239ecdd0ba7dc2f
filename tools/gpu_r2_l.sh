#!/bin/bash
# where the drop-in binary's wall time goes (ABG_HOST_TIMING), then configs[2] with the allocations logged
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2l; mkdir -p $O
W=/tmp/e2e; rm -rf $W; mkdir -p $W; cd $W
python - <<PY
import sys
sys.path.insert(0, "$R")
from abyss_amd import synth
m1, m2 = synth.make_read_set(30000000, 50.0)
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
PY
ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> $O/host_timing.err; grep -c '>' out.fa; grep "host " $O/host_timing.err
cd $R
ABG_MEM_DEBUG=1 timeout 1500 python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_config2.json 2> $O/bench_config2.err; cut -c1-700 $O/bench_config2.json; echo; grep -v "^\[mem\]" $O/bench_config2.err | tail -3; grep "^\[mem\]" $O/bench_config2.err | tail -40
