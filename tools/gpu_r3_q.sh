#!/bin/bash
# classification certificates (CertDB): on / off, configs[1]
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3q; mkdir -p $O
cd $R
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-10s %.0f Mk/s %.1f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]), d["pass_ms_per_step"], "parity", d["parity"]["ok"], "cert_hits", d["engine_stats"].get("cert_hits"), {k: round(v["ms"],1) for k, v in d["kernel_ms"].items() if k in ("classify","rewalk","cert_append","reclassify","presearch","contig_prep")})
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
run cert1 ABG_CERT=1
run cert0 ABG_CERT=0
run cert1_l22 ABG_CERT=1 ABG_CERT_LOG2=22
run cert1_b ABG_CERT=1
