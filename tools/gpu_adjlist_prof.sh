#!/bin/bash
# kernel trace of the drop-in AdjList on the unitigs of a 1 M-pair read set (rocprofv3 --kernel-trace --stats)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/adjprof; mkdir -p $O
mkdir -p /tmp/adjp && cd /tmp/adjp && export TMPDIR=/tmp
python - <<PY
import sys
sys.path.insert(0, "$R")
from abyss_amd import synth
m1, m2 = synth.make_read_set(6_000_000, 50.0)
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
PY
$R/abyss_amd/bin/abyss-bloom-dbg -k64 -b512M -H4 -q3 -j64 r1.fq r2.fq > unitigs-1.fa
grep -c ">" unitigs-1.fa
t0=$(date +%s%N); ABG_ADJ_TIMING=1 $R/abyss_amd/bin/AdjList -k64 -m50 --dot unitigs-1.fa > u.dot 2> $O/timing.txt; t1=$(date +%s%N)
echo "AdjList wall $(( (t1 - t0) / 1000000 )) ms, $(grep -c ' -> ' u.dot) edges"; cat $O/timing.txt
rm -rf /tmp/adjp/prof
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/adjp/prof -o adj -- $R/abyss_amd/bin/AdjList -k64 -m50 --dot unitigs-1.fa > u2.dot 2> /dev/null
cmp u.dot u2.dot && echo same-output
find /tmp/adjp/prof -name "*kernel_stats.csv" -exec cp {} $O/adjlist_kernel_stats.csv \;
cat $O/adjlist_kernel_stats.csv | cut -c1-160
