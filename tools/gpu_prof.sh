#!/bin/bash
# rocprofv3 runs on the GPU box: kernel trace + stats of the bench, and PMC passes.
# Raw output stays in /tmp; only the small summaries are copied to gpurun_out/.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
MODE=${1:-all}
if [ "$MODE" = "all" ] || [ "$MODE" = "trace" ]; then
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o trace -- python $R/bench.py --warmup 0 --steps 1 --no-cpu-baseline > /tmp/prof_trace.log 2>&1
tail -3 /tmp/prof_trace.log | cut -c1-400
find /tmp/prof_trace -name "*stats*.csv" -exec cp {} $R/gpurun_out/prof/ \;
fi
if [ "$MODE" = "all" ] || [ "$MODE" = "pmc" ]; then
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d /tmp/prof_pmc_$i -o pmc -- python $R/bench.py --pairs 500000 --bloom 256M --warmup 0 --steps 1 --no-cpu-baseline > /tmp/prof_pmc_$i.log 2>&1
  tail -2 /tmp/prof_pmc_$i.log | cut -c1-300
done
fi
cd $R
python - <<'PY'
import csv, glob, collections
for f in glob.glob('/tmp/prof_trace/**/*kernel_stats.csv', recursive=True):
    print('==', f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 16: print(','.join(row)[:220])
for d in sorted(glob.glob('/tmp/prof_pmc_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for row in csv.DictReader(open(f)):
            name = row.get('Kernel_Name', '')[:70]
            agg[name][row['Counter_Name']] += float(row['Counter_Value'])
        out = open('gpurun_out/prof/pmc_%s.txt' % d[-1], 'w')
        for name, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
            line = name + ' ' + ' '.join('%s=%.4g' % kv for kv in sorted(c.items()))
            out.write(line + '\n')
        print('==', f)
        for name, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:7]:
            print(name, ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
PY
du -sh gpurun_out
