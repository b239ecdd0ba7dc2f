#!/bin/bash
# where abg_assemble_seqs_v's host time goes on the full configs[1] read set
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p; mkdir -p $O
W=/tmp/e2e; rm -rf $W; mkdir -p $W; cd $W
python - <<PY
import sys
sys.path.insert(0, "$R")
from abyss_amd import synth
m1, m2 = synth.make_read_set(30000000, 50.0)
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
PY
for i in 1 2; do ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> $O/host_timing.err; grep "host" $O/host_timing.err | tail -5; done
