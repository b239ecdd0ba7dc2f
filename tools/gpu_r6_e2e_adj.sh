# AdjList and abyss-rresolver-short on configs[1]'s unitigs, process start to output written: contigs parsed block-parallel (default) / by one thread (ABG_FASTA_BLOCKS_MIN=-1)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/e2e && cd /tmp/e2e
python - <<PY
import sys
sys.path.insert(0, "$R")
from abyss_amd import synth
h1, h2 = synth.make_genome(30_000_000, seed=42)
m1, m2 = synth.sample_pairs_cb(h1, h2, 5_000_000, read_len=150, err=0.005, seed=7)
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
PY
sync
$R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > unitigs-1.fa
for run in 1 2 3; do
for v in 1048576 -1; do
  sleep 2
  t0=$(date +%s%N)
  ABG_FASTA_BLOCKS_MIN=$v $R/abyss_amd/bin/AdjList -k64 -m50 --dot unitigs-1.fa > unitigs-1.dot
  t1=$(date +%s%N)
  echo "AdjList blocks_min=$v wall $(( (t1 - t0) / 1000000 )) ms $(sha256sum unitigs-1.dot | cut -c1-16)"
done; done
for run in 1 2 3; do
  sleep 2
  t0=$(date +%s%N)
  ABG_RR_TIMING=1 $R/abyss_amd/bin/abyss-rresolver-short -b2G -f0.8 -j$(nproc) -k64 -h rr --dot -c rr.fa -g rr.dot unitigs-1.fa unitigs-1.dot r1.fq r2.fq > /dev/null 2> rr_err.txt
  t1=$(date +%s%N)
  echo "rresolver wall $(( (t1 - t0) / 1000000 )) ms $(sha256sum rr.fa | cut -c1-16)"
done
