#!/bin/bash
# prefix-XOR stretch hashes in the bulk steps: parity first, then the bench (with events: per-kernel numbers)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3 | cut -c1-300
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernel_ms"]
    g=lambda n: "%.0f" % k[n]["ms"] if n in k else "-"
    print("%-10s %.0f Mk/s (no events %.0f) %.1f ms/step" % (sys.argv[2], d["value"], d["no_events"]["value"], d["ms_per_step"]), d["pass_ms_per_step"], "staged", g("hash_bin_staged"), "guide", g("guide_build"), "scan", g("presearch_scan"), "presearch", g("presearch"), "classify", g("classify"), "rewalk", g("rewalk"), "parity", d["parity"]["ok"])
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
run base ABG_X=0
run base_b ABG_X=0
