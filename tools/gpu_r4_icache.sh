#!/bin/bash
# Why a trueBranch node takes ~7 us: the walkers' own breakdown (ABG_WALK_DEBUG) and the instruction cache's counters for the
# walker and pre-search kernels.  -> gpurun_out/r4i/
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4i
mkdir -p $O
export TMPDIR=/tmp
cd $R
ABG_WALK_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/walkdbg.json 2> $O/walkdbg.err; grep walkdbg $O/walkdbg.err | cut -c1-420 > $O/walkdbg.txt; head -8 $O/walkdbg.txt
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*" | sort -u | tr '\n' ' ' | cut -c1-1500; echo
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$tag -o pmc -- python $R/bench.py --warmup 0 --steps 1 --no-cpu-baseline --no-end-to-end --no-events > /tmp/pmc_$tag.log 2>&1
  tail -2 /tmp/pmc_$tag.log | cut -c1-200
done
cd $R
python - <<'PY' | tee $O/pmc_icache.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('/tmp/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        n = row['Kernel_Name']
        key = 'k_walkers<FWalk>' if 'FWalk' in n else 'k_walkers<FPresearch>' if 'FPresearch<' in n else 'FClassify' if 'FClassify' in n else 'FTilePurity' if 'FTilePurity' in n else None
        if key: agg[key][row['Counter_Name']] += float(row['Counter_Value'])
for k, c in sorted(agg.items()):
    print(k, ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
    if c.get('SQC_ICACHE_REQ'): print('   icache: hit rate %.4f, misses per 1000 wave-instructions %.2f' % (c.get('SQC_ICACHE_HITS', 0) / c['SQC_ICACHE_REQ'], 1000 * c.get('SQC_ICACHE_MISSES', 0) / max(c.get('SQ_INSTS_SALU', 0) + c.get('SQ_INSTS_VALU', 0), 1)))
PY
