#!/bin/bash
set -u
O=gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
print(sys.argv[2], "ms/step %.0f" % d["ms_per_step"], d["pass_ms_per_step"], "rewalk %.0f/%d" % (k["rewalk"]["ms"], k["rewalk"]["launches"]), "classify %.0f" % k["classify"]["ms"], "cand", s["candidates"], "unitigs", d["config"]["unitigs"])
PY
}
run base A=1
run noprefetch ABG_PREFETCH=0
run slots6144 ABG_WALK_SLOTS=6144
run slots3072 ABG_WALK_SLOTS=3072
