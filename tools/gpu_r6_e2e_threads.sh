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
for run in 1 2 3; do
for j in ${JS:-32 24 16 12 8}; do
  t0=$(date +%s%N)
  ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$j r1.fq r2.fq > out.fa 2> err.txt
  t1=$(date +%s%N)
  dev=$(grep "\[host\] load:" err.txt | sed 's/.*device \([0-9.]*\) s/\1/' | paste -sd+ | python3 -c "print(round(eval(input()),3))")
  pack=$(grep "\[host\] load:" err.txt | sed 's/.*pack \([0-9.]*\) s.*/\1/' | paste -sd+ | python3 -c "print(round(eval(input()),3))")
  echo "j=$j wall $(( (t1 - t0) / 1000000 )) ms; device sum $dev s; pack sum $pack s; $(grep 'kept reads assembled' err.txt | cut -c1-110)"
done; done
