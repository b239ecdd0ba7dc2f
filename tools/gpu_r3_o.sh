#!/bin/bash
# walker waves per SIMD after the ALU cuts (variants in abyss_amd/lib/variants)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o; mkdir -p $O
cd $R
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-events > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-14s %.0f Mk/s %.1f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]), d["pass_ms_per_step"], "parity", d["parity"]["ok"])
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
run base ABG_X=0
for v in w2 w4 w5; do run var_$v ABG_LIB=$R/abyss_amd/lib/variants/lib_$v.so; done
run base_b ABG_X=0
