#!/bin/bash
# diagnosis at scale: ABG_MEMO_VERIFY on configs[1] with the pre-search on; test_gpu_scale with it
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
ABG_PRESEARCH=1 ABG_MEMO_VERIFY=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/verify.json 2> $O/verify.err
grep -i "verify\|abyss_amd" $O/verify.err | head; python -c "
import json; d=json.load(open('$O/verify.json')); print(d['config']['unitigs'], d['engine_stats'])"
ABG_PRESEARCH=1 timeout 400 python -m pytest tests/test_gpu_scale.py -x -q 2>&1 | tail -15 | cut -c1-300
