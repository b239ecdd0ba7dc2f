# the drop-in abyss-bloom-dbg on the full configs[1] files with ABG_HOST_TIMING: every load call's and every device stage's parts (RUNS runs, 3 s apart)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r6mk}; mkdir -p $O
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
for run in $(seq 1 ${RUNS:-3}); do
  sleep 3
  t0=$(date +%s%N)
  ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> $O/err_$run.txt
  t1=$(date +%s%N)
  echo "wall $(( (t1 - t0) / 1000000 )) ms; $(grep 'kept reads assembled' $O/err_$run.txt | cut -c1-110) $(sha256sum out.fa | cut -c1-16)"
done
grep "host" $O/err_2.txt | cut -c1-200
# ... and the two rules after it
cp out.fa unitigs-1.fa
$R/abyss_amd/bin/AdjList -k64 -m50 --dot unitigs-1.fa > unitigs-1.dot
for run in 1 2 3; do
  sleep 2
  t0=$(date +%s%N)
  ABG_RR_TIMING=1 $R/abyss_amd/bin/abyss-rresolver-short -b2G -f0.8 -j$(nproc) -k64 -h rr --dot -c rr.fa -g rr.dot unitigs-1.fa unitigs-1.dot r1.fq r2.fq > /dev/null 2> $O/rr_$run.txt
  t1=$(date +%s%N)
  echo "rresolver wall $(( (t1 - t0) / 1000000 )) ms $(sha256sum rr.fa | cut -c1-16)"
done
cut -c1-200 $O/rr_3.txt | head -60
