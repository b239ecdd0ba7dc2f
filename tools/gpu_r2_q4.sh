#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2q; mkdir -p $O
cd $R
for v in 1 0 1 0; do
ABG_STAGE_HASH_EARLY=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_he$v.json 2> $O/bench_he$v.err
python - $O/bench_he$v.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]
print("hash early", sys.argv[2], "ms/step %.1f" % d["ms_per_step"], d["pass_ms_per_step"], {n:round(v["ms"],1) for n,v in k.items() if n in ("hash_staged","hash_bin_staged","tile_purity","op_target","tile_apply","insert_retry")})
PY
done
