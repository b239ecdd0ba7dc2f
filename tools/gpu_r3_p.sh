#!/bin/bash
# PASS 2 batch schedule now that the launches are bounded by their heaviest walker: fewer, larger batches?
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p; mkdir -p $O
cd $R
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-events > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); s=d["engine_stats"]; print("%-16s %.0f Mk/s %.1f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]), d["pass_ms_per_step"], "rounds", s["walk_rounds"], "cand", s["candidates"], "parity", d["parity"]["ok"])
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
run base ABG_X=0
run g3 ABG_P2_GROWTH=3
run g4 ABG_P2_GROWTH=4
run f64k ABG_P2_FIRST_BATCH=65536
run f128k ABG_P2_FIRST_BATCH=131072
run f128k_g3 ABG_P2_FIRST_BATCH=131072 ABG_P2_GROWTH=3
run f256k_g4 ABG_P2_FIRST_BATCH=262144 ABG_P2_GROWTH=4
run f512k_g4 ABG_P2_FIRST_BATCH=524288 ABG_P2_GROWTH=4
run base_b ABG_X=0
