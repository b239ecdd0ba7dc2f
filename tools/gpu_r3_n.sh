#!/bin/bash
# where the walkers' time goes now (ABG_WALK_DEBUG)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
ABG_WALK_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/dbg.json 2> $O/dbg.err
grep walkdbg $O/dbg.err > $O/walkdbg.txt
python - $O/walkdbg.txt <<'PY'
import re,sys
L=open(sys.argv[1]).read().split('\n')
n=None; tot=[0]*6
for l in L:
    m=re.search(r'rewalk n=(\d+)',l)
    if m: n=int(m.group(1))
    m=re.search(r'sum: t=([\d.]+)ms \(search ([\d.]+) \[chains ([\d.]+)\] in (\d+) calls, (\d+) tbnodes; linear ([\d.]+) of which bulk ([\d.]+).*post ([\d.]+)\) steps=(\d+)',l)
    if m:
        t=float(m.group(1)); s=float(m.group(2)); li=float(m.group(6)); b=float(m.group(7)); po=float(m.group(8))
        print("  n=%6d sum %.0f ms (/3072 = %.1f) search %.0f lin %.0f (bulk %.0f) post %.0f other %.0f" % (n,t,t/3072,s,li,b,po,t-s-li-po))
        for i,v in enumerate((t,s,li,b,po,t-s-li-po)): tot[i]+=v
    m=re.search(r'bulk examine phase ([\d.]+)',l)
    if m: print("      examine %.0f" % float(m.group(1)))
print("TOTAL t %.0f search %.0f lin %.0f bulk %.0f post %.0f other %.0f" % tuple(tot))
for l in L:
    if 'slowest:' in l: print(l[:200])
PY
