#!/bin/bash
# End-of-round evidence on a gpurun box: gpu tests, smoke, the default bench line, the rocprofv3 kernel
# trace (+ timeline of PASS 2) of the same workload, the TCC traffic counters and the walkers' SQ counters.
# Summaries -> gpurun_out/final/ (copy what is to be judged into profiles/).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
# the partitioned code path on one rank through RCCL (every collective an identity): what that path costs by itself
ABG_FORCE_DIST=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_forced_partitioned_1rank.json 2> $O/bench_forced.err; cut -c1-300 $O/bench_forced_partitioned_1rank.json
# configs[3] (spaced seed) and its parity check
timeout 300 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-300 $O/bench_config3.json
# kernel trace + timeline (tools/gpu_r3_timeline.sh writes gpurun_out/r3t/)
bash tools/gpu_r3_timeline.sh 1 > $O/timeline.log 2>&1; tail -5 $O/timeline.log | cut -c1-200
cp gpurun_out/r3t/timeline_ps1.txt $O/timeline_config1.txt 2>/dev/null; cp gpurun_out/r3t/kernel_stats_ps1.csv $O/kernel_stats_config1.csv 2>/dev/null
# HBM-side traffic (TCC) per kernel, one counter per pass
bash tools/gpu_pmc_traffic.sh --warmup 0 --steps 1 --no-cpu-baseline --no-end-to-end --no-events > $O/pmc.log 2>&1; cp gpurun_out/prof/pmc_traffic.json $O/ 2>/dev/null; tail -3 $O/pmc.log | cut -c1-200
# what the walker waves spend their cycles on
cd /tmp
rm -rf /tmp/pmc_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d /tmp/pmc_sq -o pmc -- python $R/bench.py --warmup 0 --steps 1 --no-cpu-baseline --no-end-to-end --no-events > /tmp/pmc_sq.log 2>&1
cd $R
python - <<'PY' > $O/pmc_sq_walkers.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('/tmp/pmc_sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name']
        key = 'k_walkers<FWalk>' if 'FWalk' in name else 'k_walkers<FPresearch>' if 'FPresearch<' in name else 'FClassify' if 'FClassify' in name else None
        if key: agg[key][row['Counter_Name']] += float(row['Counter_Value'])
for k, c in agg.items():
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    print(k, ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
    print('   fractions of wave cycles: active %.3f (valu %.3f, scalar %.3f), wait_any %.3f, wait_inst %.3f' % (
        c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc, c.get('SQ_ACTIVE_INST_SCA', 0) / wc, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc))
PY
cat $O/pmc_sq_walkers.txt | cut -c1-300
