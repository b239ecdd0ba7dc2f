#!/bin/bash
# diagnosis of the pre-search's wrong answers: who wrote / who checked, self-check, fewer slots
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  echo "== $tag: $(grep -i "MEMO_VERIFY" $O/$tag.err | head -3)"
  python -c "
import json; d=json.load(open('$O/$tag.json')); s=d['engine_stats']; print('   unitigs', d['config']['unitigs'], 'ms', round(d['ms_per_step']), 'pre', s['pre_requests'], s['pre_adds'], 'memo', s['memo_hits'], s['memo_adds'], 'presearch ms', d['kernel_ms'].get('presearch'))"
}
run walkers_verify ABG_MEMO_VERIFY=1
run presearch_selfverify ABG_MEMO_VERIFY=2
run slots64 ABG_PRESEARCH_SLOTS=64
run slots64_verify ABG_PRESEARCH_SLOTS=64 ABG_MEMO_VERIFY=1
run noguide_verify ABG_GUIDE_STRIDE=0 ABG_MEMO_VERIFY=1
