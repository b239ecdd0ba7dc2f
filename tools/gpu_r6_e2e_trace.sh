#!/bin/bash
# kernel trace of the drop-in binary on the full configs[1] FASTQ files: how busy the GPU is while the host parses (where end-to-end time goes)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r6e2e}; mkdir -p $O
mkdir -p /tmp/e2e && cd /tmp/e2e
python - <<PY
import sys, time
sys.path.insert(0, "$R")
from abyss_amd import synth
h1, h2 = synth.make_genome(30_000_000, seed=42)
m1, m2 = synth.sample_pairs_cb(h1, h2, 5_000_000, read_len=150, err=0.005, seed=7)
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
PY
export TMPDIR=/tmp
ABG_HOST_TIMING=1 $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> $O/warm.txt
rm -rf /tmp/e2e_tr
ABG_HOST_TIMING=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/e2e_tr -o tr -- $R/abyss_amd/bin/abyss-bloom-dbg -k64 -b2G -H4 -q3 -j$(nproc) r1.fq r2.fq > out.fa 2> $O/traced.txt
grep "host" $O/traced.txt | grep -v "load:" | cut -c1-160
python - <<'PY' > $O/summary.txt
import csv, glob, re, collections
import os
fs = glob.glob('/tmp/e2e_tr/**/*kernel_trace.csv', recursive=True)
if not fs: print('no kernel trace:', [os.path.join(d, x) for d, _, xs in os.walk('/tmp/e2e_tr') for x in xs][:20]); raise SystemExit
f = fs[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    m = re.search(r'(k_\w+)<.*?abg::(\w+)', n)
    if m: return m.group(2)
    m = re.search(r'(k_\w+)', n)
    return m.group(1) if m else n[:30]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
t0 = ev[0][0]
g = next(i for i, e in enumerate(ev) if 'FGuideBuild' in e[2])
for name, seg in (("PASS 1 (first kernel .. guide build)", ev[:g]), ("PASS 2", ev[g:])):
    a, b = seg[0][0], max(e[1] for e in seg)
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, n in seg[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = collections.Counter()
    for s, e, n in seg: tot[n] += e - s
    print("%s: wall %.1f ms, some kernel running %.1f ms, kernels summed %.1f ms, %d launches" % (name, (b - a) / 1e6, busy / 1e6, sum(tot.values()) / 1e6, len(seg)))
    print("   " + ", ".join("%s %.1f" % (k, v / 1e6) for k, v in tot.most_common(14)))
    # gaps > 2 ms
    gaps = []; end = seg[0][1]
    for s, e, n in seg[1:]:
        if s - end > 2e6: gaps.append(((end - a) / 1e6, (s - end) / 1e6, n))
        end = max(end, e)
    print("   gaps > 2 ms (at ms, length, next kernel): " + "; ".join("%.0f +%.1f %s" % g for g in gaps[:40]))
PY
cat $O/summary.txt | cut -c1-1500
