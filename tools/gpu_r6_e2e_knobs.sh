# the drop-in binaries on the full configs[1] files with a knob varied: KNOB=name VALUES="a b c" (bloom-dbg), RRKNOB / RRVALUES (rresolver); three runs each, interleaved
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
for run in 1 2 3; do
for v in ${VALUES:-default}; do
  sleep 3
  t0=$(date +%s%N)
  if [ "$v" = default ]; then ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> err.txt
  else env ${KNOB}=$v ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> err.txt; fi
  t1=$(date +%s%N)
  echo "${KNOB:-}=$v wall $(( (t1 - t0) / 1000000 )) ms; created $(grep 'context created' err.txt | cut -c7-13) last chunk $(grep 'chunk loaded' err.txt | tail -1 | cut -c7-13) $(grep 'kept reads assembled' err.txt | cut -c1-110) $(sha256sum out.fa | cut -c1-16)"
done; done
if [ -n "${RRVALUES:-}" ]; then
cp out.fa unitigs-1.fa
$R/abyss_amd/bin/AdjList -k64 -m50 --dot unitigs-1.fa > unitigs-1.dot
for run in 1 2 3; do
for v in $RRVALUES; do
  sleep 2
  t0=$(date +%s%N)
  env ${RRKNOB}=$v ABG_RR_TIMING=1 $R/abyss_amd/bin/abyss-rresolver-short -b2G -f0.8 -j$(nproc) -k64 -h rr --dot -c rr.fa -g rr.dot unitigs-1.fa unitigs-1.dot r1.fq r2.fq > /dev/null 2> rr_err.txt
  t1=$(date +%s%N)
  echo "rresolver ${RRKNOB}=$v wall $(( (t1 - t0) / 1000000 )) ms $(sha256sum rr.fa | cut -c1-16) $(grep -h 'filter built\|contigs read' rr_err.txt | tr -s ' ' | tr '\n' ';')"
done; done
fi
