#!/bin/bash
# round 2 profile of the default bench command: rocprofv3 kernel trace (+ stats, + how much of the
# step the GPU is busy) and the TCC HBM-traffic counters (separate passes).  usage: gpu_r2_prof.sh TAG [trace|pmc|all]
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}; MODE=${2:-all}
O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--warmup 1 --steps 1 --no-cpu-baseline"
if [ "$MODE" = "all" ] || [ "$MODE" = "trace" ]; then
rm -rf /tmp/prof_trace
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o trace -- python $R/bench.py $ARGS > $O/bench_under_trace.json 2> /tmp/prof_trace.err
find /tmp/prof_trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob('/tmp/prof_trace/**/*kernel_trace.csv', recursive=True):
    ev = []
    for row in csv.DictReader(open(f)):
        ev.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), row['Kernel_Name']))
    ev.sort()
    # the last step = the second half of the trace by time (warm-up 1 + 1 step): take kernels after the midpoint gap
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    open(O + '/gpu_busy.txt', 'w').write("kernels %d span %.1f ms busy(union) %.1f ms idle %.1f ms\n" % (len(ev), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    print(open(O + '/gpu_busy.txt').read())
PY
head -25 $O/kernel_stats.csv | cut -c1-200
fi
if [ "$MODE" = "all" ] || [ "$MODE" = "pmc" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 1500 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --warmup 0 --steps 1 --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  tail -1 /tmp/pmc_$c.log | cut -c1-120
done
python - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
keys = ("FHashOps", "FBinCoarse", "FBinFine", "FTilePurity", "FOpTarget", "FTileApply", "FClaimList", "FInsertRound", "k_insert_drain",
        "FHashClaim", "FClassify", "FRefilter", "k_walkers", "FGuideBuild", "FContigPrep", "FReadPrep", "FPreCommit", "FPcTimeMin", "FPcDecide", "FPcApply", "DeviceSelect", "DeviceCompact")
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != c: continue
            name = row['Kernel_Name']
            for key in keys:
                if key in name:
                    agg[key][0] += float(row['Counter_Value']); agg[key][1] += 1
                    break
            else:
                agg["other"][0] += float(row['Counter_Value']); agg["other"][1] += 1
        for k, (v, n) in agg.items():
            out.setdefault(k, {})[c] = {"sum": v, "dispatches": n}
out["_note"] = "TCC counters of one step of `bench.py --warmup 0 --steps 1`, summed over all dispatches of a kernel; KB units (MI355X_MICROARCH.md)"
json.dump(out, open(O + '/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    if k[0] != '_': print(k, {c: round(x["sum"] * 1024 / 1e9, 1) for c, x in v.items()}, "GB")
PY
fi
