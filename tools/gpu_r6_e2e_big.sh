# the drop-in abyss-bloom-dbg on a read set PAIRS (default 20 M) pairs large -- four times configs[1]'s files, ~30 load calls -- with this round's host paths
# (block parser, uploads ahead) and without them: the two FASTA files must be equal
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r6big}; mkdir -p $O
mkdir -p /tmp/e2e && cd /tmp/e2e
python - <<PY
import sys
sys.path.insert(0, "$R")
from abyss_amd import synth
h1, h2 = synth.make_genome(${GENOME:-120000000}, seed=43)
m1, m2 = synth.sample_pairs_cb(h1, h2, ${PAIRS:-20000000}, read_len=150, err=0.005, seed=8)
synth.write_fastq("b1.fq", m1, "r", 1); synth.write_fastq("b2.fq", m2, "r", 2)
PY
sync; ls -la b1.fq b2.fq
for mode in new old new; do
  sleep 3
  t0=$(date +%s%N)
  if [ $mode = new ]; then ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b8G -H4 -q3 -j$(nproc) b1.fq b2.fq > out_$mode.fa 2> $O/err_$mode.txt
  else ABG_NO_UPLOAD_AHEAD=1 ABG_READER_FAST=0 ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b8G -H4 -q3 -j$(nproc) b1.fq b2.fq > out_$mode.fa 2> $O/err_$mode.txt; fi
  rc=$?
  t1=$(date +%s%N)
  echo "$mode rc=$rc wall $(( (t1 - t0) / 1000000 )) ms; $(grep -c 'chunk loaded' $O/err_$mode.txt) load calls; $(grep 'kept reads assembled' $O/err_$mode.txt | cut -c1-110) $(grep -c '>' out_$mode.fa) unitigs $(sha256sum out_$mode.fa | cut -c1-16)"
done
