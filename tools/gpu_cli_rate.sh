#!/bin/bash
# PCIe-inclusive rate of the drop-in binary: FASTQ files -> unitig FASTA, wall clock.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/clirate && cd /tmp/clirate
python - <<PY
import sys, time
sys.path.insert(0, "$R")
from abyss_amd import synth
t=time.time()
m1, m2 = synth.make_read_set(6_000_000, 50.0)   # 1 M pairs
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
print("reads", m1.shape, "written in %.1f s" % (time.time()-t))
PY
ls -la r1.fq
t0=$(date +%s%N)
$R/abyss_amd/bin/abyss-bloom-dbg -k64 -b512M -H4 -q3 -j1 -v r1.fq r2.fq > out.fa 2> err.txt
t1=$(date +%s%N)
echo "wall $(( (t1 - t0) / 1000000 )) ms for 174000000 read k-mers"; grep -c ">" out.fa; tail -3 err.txt | head -2
