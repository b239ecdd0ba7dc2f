#!/bin/bash
# kernel timeline of one configs[1] step (rocprofv3 --kernel-trace): where PASS 2's time goes between the launches
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r3t}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${1:-1}; do
rm -rf /tmp/tl_$v
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl_$v -o tl -- python $R/bench.py --warmup 1 --steps 1 --no-cpu-baseline --no-events > /tmp/tl_$v.log 2>&1
tail -1 /tmp/tl_$v.log | cut -c1-300
find /tmp/tl_$v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_ps$v.csv \;
python - $v <<'PY' > $O/timeline_ps$v.txt
import csv, glob, sys, re
f = glob.glob('/tmp/tl_%s/**/*kernel_trace.csv' % sys.argv[1], recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    m = re.search(r'(k_\w+)<.*?abg::(\w+)', n)
    if m: return m.group(1) + ':' + m.group(2)
    m = re.search(r'(k_\w+)', n)
    return m.group(1) if m else n[:40]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
ev.sort()
# the last step: from the last FHashOps burst... take events after the last 'FKmerCounts' occurrence
idx = max(i for i, e in enumerate(ev) if 'FKmerCounts' in e[2])
ev = ev[idx:]
t0 = ev[0][0]
# PASS 2 starts at the guide build
g = next(i for i, e in enumerate(ev) if 'FGuideBuild' in e[2])
print('step total %.1f ms; pass1 %.1f ms; pass2 %.1f ms' % ((ev[-1][1] - t0) / 1e6, (ev[g][0] - t0) / 1e6, (ev[-1][1] - ev[g][0]) / 1e6))
p2 = ev[g:]
base = p2[0][0]
busy_end = base
print('PASS 2 timeline (ms from its start): kernels >= 0.3 ms, and gaps >= 0.3 ms with nothing running')
run_end = base
for s, e, n, q in p2:
    if s > run_end and (s - run_end) / 1e6 >= 0.3:
        print('   %8.2f  gap %.2f' % ((run_end - base) / 1e6, (s - run_end) / 1e6))
    if (e - s) / 1e6 >= 0.3:
        print('%8.2f %8.2f  %-28s q%s' % ((s - base) / 1e6, (e - s) / 1e6, n, q))
    run_end = max(run_end, e)
# totals by kernel over PASS 2
tot = {}
for s, e, n, q in p2: tot[n] = tot.get(n, 0) + (e - s) / 1e6
print('totals:', ', '.join('%s %.1f' % kv for kv in sorted(tot.items(), key=lambda kv: -kv[1])[:16]))
gaps = 0; run_end = base
for s, e, n, q in p2:
    if s > run_end: gaps += s - run_end
    run_end = max(run_end, e)
print('idle (no kernel running) in PASS 2: %.1f ms' % (gaps / 1e6))
# PASS 1: what its wall time is made of -- busy time per stream, time with nothing running anywhere,
# and the time the MAIN stream (the one with the tile kernels) has nothing running
p1 = ev[:g]
tile_q = next((q for s, e, n, q in p1 if 'FTileApply' in n), None)
tot1 = {}
for s, e, n, q in p1: tot1[n] = tot1.get(n, 0) + (e - s) / 1e6
print('PASS 1 totals:', ', '.join('%s %.1f' % kv for kv in sorted(tot1.items(), key=lambda kv: -kv[1])[:14]))
idle_all = 0; run_end = p1[0][0]
for s, e, n, q in p1:
    if s > run_end: idle_all += s - run_end
    run_end = max(run_end, e)
main = [(s, e, n) for s, e, n, q in p1 if q == tile_q]
idle_main = 0; gaps_main = []; run_end = main[0][0]
for s, e, n in main:
    if s > run_end: idle_main += s - run_end; gaps_main.append(((s - run_end) / 1e6, n))
    run_end = max(run_end, e)
busy_main = sum(e - s for s, e, n in main) / 1e6
print('PASS 1: %.1f ms; main stream busy %.1f ms in %d kernels, idle %.1f ms in %d gaps; nothing running on any stream %.1f ms' % (
    (p1[-1][1] - p1[0][0]) / 1e6, busy_main, len(main), idle_main / 1e6, len(gaps_main), idle_all / 1e6))
by = {}
for d, n in gaps_main: by[n] = (by.get(n, (0, 0))[0] + d, by.get(n, (0, 0))[1] + 1)
print('main-stream gaps by the kernel that follows:', ', '.join('%s %.1f ms / %d' % (n, v[0], v[1]) for n, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:10]))
PY
done
tail -n 6 $O/timeline_ps${1:-1}.txt | cut -c1-600
