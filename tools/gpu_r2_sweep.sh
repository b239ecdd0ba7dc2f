#!/bin/bash
set -u
O=gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
print(sys.argv[2], "ms/step %.0f" % d["ms_per_step"], d["pass_ms_per_step"], "rewalk %.0f/%d" % (k["rewalk"]["ms"], k["rewalk"]["launches"]), "cand", s["candidates"], "unitigs", d["config"]["unitigs"])
PY
}
run f32k_m4m ABG_P2_MAX_BATCH=4194304 ABG_P2_FIRST_BATCH=32768
run f32k_m8m ABG_P2_MAX_BATCH=8388608 ABG_P2_FIRST_BATCH=32768
run f32k_m16m ABG_P2_MAX_BATCH=16777216 ABG_P2_FIRST_BATCH=32768
run f24k_m8m ABG_P2_MAX_BATCH=8388608 ABG_P2_FIRST_BATCH=24576
run f32k_m8m_g3 ABG_P2_MAX_BATCH=8388608 ABG_P2_FIRST_BATCH=32768 ABG_P2_GROWTH=3
