#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2q; mkdir -p $O
cd $R
for v in late early late early; do
if [ $v = early ]; then export ABG_STAGE_EARLY=1; else unset ABG_STAGE_EARLY; fi
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
python - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]
print("stage", sys.argv[2], "ms/step %.1f" % d["ms_per_step"], d["pass_ms_per_step"], {n:round(v["ms"],1) for n,v in k.items() if n in ("hash_bin_staged","tile_purity","op_target","tile_apply","insert_retry")})
PY
done
