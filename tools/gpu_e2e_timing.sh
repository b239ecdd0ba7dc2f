#!/bin/bash
# where the drop-in binary's wall time goes on the full configs[1] FASTQ files (ABG_HOST_TIMING marks)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e2e_t; mkdir -p $O
mkdir -p /tmp/e2e && cd /tmp/e2e
python - <<PY
import sys, time
sys.path.insert(0, "$R")
from abyss_amd import synth
t = time.time()
h1, h2 = synth.make_genome(30_000_000, seed=42)
m1, m2 = synth.sample_pairs_cb(h1, h2, 5_000_000, read_len=150, err=0.005, seed=7)
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
print("files written in %.1f s" % (time.time() - t))
PY
for run in 1 2 3; do
  for extra in "${@:-ABG_X=0}"; do
    t0=$(date +%s%N)
    env $extra ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> $O/err_${run}_${extra%%=*}.txt
    t1=$(date +%s%N)
    echo "run $run [$extra]: wall $(( (t1 - t0) / 1000000 )) ms; sha256 $(sha256sum out.fa | cut -c1-16)"
    grep "host" $O/err_${run}_${extra%%=*}.txt | cut -c1-220
  done
done
