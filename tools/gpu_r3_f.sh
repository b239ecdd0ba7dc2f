#!/bin/bash
# solid bit plane (ABG_SOLID_PLANE) A/B with and without the pre-search, without per-launch events; parity subset first
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
for cfg in "1 1" "0 1" "1 0" "0 0"; do
set -- $cfg
ABG_SOLID_PLANE=$1 ABG_PRESEARCH=$2 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_sp$1_ps$2.json 2> $O/bench_sp$1_ps$2.err
python - $O/bench_sp$1_ps$2.json "plane=$1 presearch=$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
    g=lambda n: "%.0f/%d" % (k[n]["ms"], k[n]["launches"]) if n in k else "-"
    print(sys.argv[2], "Mk/s %.0f (no events %.0f) ms/step %.1f" % (d["value"], d.get("no_events",{}).get("value",0), d["ms_per_step"]), d["pass_ms_per_step"], "rewalk", g("rewalk"), "presearch", g("presearch"), "classify", g("classify"), "guide", g("guide_build"), "prep", g("contig_prep"), "unitigs", d["config"]["unitigs"], d["config"]["unitig_bp"], d.get("parity"))
except Exception as e:
    print(sys.argv[2], "ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
