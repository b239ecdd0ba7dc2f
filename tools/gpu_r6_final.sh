#!/bin/bash
# Evidence at the head of round 6, in two gpurun calls (the bench line quotes the TCC traffic from the profiles/ file taken on
# exactly these kernel sources: the counters come first, are copied into profiles/, and the line is taken after):
#   COMMIT=$(git rev-parse --short HEAD) gpurun -- 'COMMIT=... bash tools/gpu_r6_final.sh counters'   -> gpurun_out/r6final/{pmc_traffic.json, pmc_sq_walkers.txt, kernel_stats.csv}
#   gpurun -- 'bash tools/gpu_r6_final.sh line'       -> gpurun_out/r6final/{pytest_gpu.log, smoke.log, bench_default.json, bench_noevents.json, e2e.txt}
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6final
mkdir -p $O
export TMPDIR=/tmp
cd $R
if [ "${1:-line}" = counters ]; then
  bash tools/gpu_pmc_traffic.sh --warmup 0 --steps 1 --no-cpu-baseline --no-end-to-end --no-events > $O/pmc.log 2>&1; cp gpurun_out/prof/pmc_traffic.json $O/ 2>/dev/null; tail -2 $O/pmc.log | cut -c1-200
  cd /tmp; rm -rf /tmp/pmc_sq
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d /tmp/pmc_sq -o pmc -- python $R/bench.py --warmup 0 --steps 1 --no-cpu-baseline --no-end-to-end --no-events > /tmp/pmc_sq.log 2>&1
  cd $R
  python - <<'PY' > $O/pmc_sq_walkers.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('/tmp/pmc_sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name']
        key = 'k_walkers<FWalk>' if 'FWalk' in name else 'FClassify' if 'FClassify' in name else 'FTileApply' if 'FTileApply' in name else None
        if key: agg[key][row['Counter_Name']] += float(row['Counter_Value'])
for k, c in agg.items():
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    print(k, ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
    print('   fractions of wave cycles: active %.3f (valu %.3f, scalar %.3f), wait_any %.3f, wait_inst %.3f' % (
        c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc, c.get('SQ_ACTIVE_INST_SCA', 0) / wc, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc))
PY
  cat $O/pmc_sq_walkers.txt | cut -c1-300
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r6f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/prof_bench.json 2> $O/prof.err
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -8 $O/kernel_stats.csv | cut -c1-150
  rm -rf $O/prof
  exit 0
fi
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-events > $O/bench_noevents.json 2> $O/bench_noevents.err
python - <<'PY'
import json
for f in ("bench_default", "bench_noevents"):
    d = json.loads([l for l in open("gpurun_out/r6final/%s.json" % f) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f, round(d["value"], 1), round(d["ms_per_step"], 1), d.get("pass_ms_per_step"), d.get("parity", {}).get("ok"), r["kernel"], r["frac"], r.get("traffic"), r.get("traffic_kernels_are_head"), (d.get("end_to_end") or {}).get("wall_ms"), (d.get("cpu_baseline") or {}).get("value"))
PY
