#!/bin/bash
# SQ counters of chosen kernels over one configs[1] step (rocprofv3 --pmc, its own run): how a kernel's wave cycles divide into
# issuing and waiting.  usage: OUT=dir KERNELS="FClassify FWalk" [env...] bash tools/gpu_r6_sq.sh
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r6sq}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d /tmp/pmc_sq -o pmc -- python $R/bench.py --warmup 0 --steps 1 --no-cpu-baseline --no-end-to-end --no-events > /tmp/pmc_sq.log 2>&1
tail -1 /tmp/pmc_sq.log | cut -c1-200
cd $R
KERNELS="${KERNELS:-FClassify FWalk FTileApply}" python - <<'PY' > $O/pmc_sq.txt
import csv, glob, collections, os
keys = os.environ["KERNELS"].split()
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('/tmp/pmc_sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name']
        key = next((k for k in keys if k in name), None)
        if key: agg[key][row['Counter_Name']] += float(row['Counter_Value'])
for k, c in agg.items():
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    print(k, ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
    print('   fractions of wave cycles: active %.3f (valu %.3f, scalar %.3f), wait_any %.3f, wait_inst %.3f' % (
        c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc, c.get('SQ_ACTIVE_INST_SCA', 0) / wc, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc))
PY
cat $O/pmc_sq.txt | cut -c1-330
