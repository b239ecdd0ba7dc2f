#!/bin/bash
# batch-schedule / guide sweep on config 1 (one timed step each, no warm-up)
set -u
O=gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
print(sys.argv[2], "ms/step %.0f" % d["ms_per_step"], "rewalk %.0f/%d" % (k["rewalk"]["ms"], k["rewalk"]["launches"]), "classify %.0f" % k["classify"]["ms"], "cand", s["candidates"], "unitigs", d["config"]["unitigs"])
PY
}
run base A=1
run first64k ABG_P2_FIRST_BATCH=65536
run first256k ABG_P2_FIRST_BATCH=262144
run first1m ABG_P2_FIRST_BATCH=1048576
run growth4 ABG_P2_GROWTH=4
run first64k_g4 ABG_P2_FIRST_BATCH=65536 ABG_P2_GROWTH=4
run stride2 ABG_GUIDE_STRIDE=2
run stride8 ABG_GUIDE_STRIDE=8
