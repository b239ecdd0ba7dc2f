# the drop-in binaries end to end on the full configs[1] files with the block parser on / off (ABG_READER_FAST), three runs each, 3 s apart
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r6rd}; mkdir -p $O
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
$R/tools/ubench/reader_bench r1.fq 32 3; $R/tools/ubench/reader_bench r1.fq 32 3; ABG_READER_FAST=0 $R/tools/ubench/reader_bench r1.fq 32 3
$R/tools/ubench/reader_bench r1.fq 64 3; $R/tools/ubench/reader_bench r1.fq 16 3
for run in 1 2 3; do
for fast in 1 0; do
  sleep 3
  t0=$(date +%s%N)
  ABG_READER_FAST=$fast ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out$fast.fa 2> $O/err_${fast}_$run.txt
  t1=$(date +%s%N)
  echo "fast=$fast wall $(( (t1 - t0) / 1000000 )) ms; $(grep 'kept reads assembled' $O/err_${fast}_$run.txt | cut -c1-110) $(sha256sum out$fast.fa | cut -c1-16)"
done; done
grep "host" $O/err_1_3.txt | cut -c1-150
