#!/bin/bash
# What the user sees on the FULL configs[1] read set (5 M x 2x150 bp, k=64, B=2G, H=4): the drop-in binary end
# to end (FASTQ in, FASTA out; plain and .gz input) next to the unmodified reference on the same files, same box.
# Writes gpurun_out/e2e/*.json.  usage: gpu_r2_e2e.sh [pairs] [ref|noref]
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e2e; mkdir -p $O
PAIRS=${1:-5000000}; REF=${2:-ref}
W=/tmp/e2e; rm -rf $W; mkdir -p $W; cd $W
python - <<PY
import sys, time
sys.path.insert(0, "$R")
from abyss_amd import synth
t = time.time()
m1, m2 = synth.make_read_set(int($PAIRS * 300 / 50), 50.0)
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
print("reads", m1.shape, "written in %.1f s" % (time.time() - t))
PY
ls -la r1.fq | awk '{print $5, "bytes per file"}'
NPROC=$(nproc)
KMERS=$(( PAIRS * 2 * 87 ))
run() { # tag, command...
  tag=$1; shift
  t0=$(date +%s%N); "$@" > $tag.fa 2> $tag.err; rc=$?; t1=$(date +%s%N)
  ms=$(( (t1 - t0) / 1000000 ))
  echo "$tag: rc=$rc wall ${ms} ms unitigs $(grep -c '>' $tag.fa) md5 $(md5sum < $tag.fa | cut -c1-12)"
  echo "{\"tag\": \"$tag\", \"rc\": $rc, \"wall_ms\": $ms, \"read_kmers\": $KMERS, \"mkmers_per_s\": $(python -c "print(round($KMERS / $ms / 1e3, 2))"), \"threads\": $NPROC, \"unitigs\": $(grep -c '>' $tag.fa), \"fasta_md5\": \"$(md5sum < $tag.fa | cut -c1-32)\"}" > $O/$tag.json
}
run amd_fastq $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$NPROC r1.fq r2.fq
( gzip -1 -k r1.fq & gzip -1 -k r2.fq & wait )
run amd_gz $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$NPROC r1.fq.gz r2.fq.gz
run amd_fastq_j1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j1 r1.fq r2.fq
if [ "$REF" = "ref" ] && [ -x $R/oracle/_ref/abyss-bloom-dbg ]; then
  export OMP_NUM_THREADS=$NPROC
  run ref_fastq_j$NPROC $R/oracle/_ref/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$NPROC r1.fq r2.fq
fi
nproc; grep -m1 "model name" /proc/cpuinfo
