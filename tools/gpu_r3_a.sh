#!/bin/bash
# round 3, first GPU call: pre-search (ABG_PRESEARCH) parity + A/B on configs[1] + per-launch walker breakdown
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
for v in 1 0; do
ABG_PRESEARCH=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_ps$v.json 2> $O/bench_ps$v.err
python - $O/bench_ps$v.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
    g=lambda n: "%.0f/%d" % (k[n]["ms"], k[n]["launches"]) if n in k else "-"
    print("presearch", sys.argv[2], "Mk/s %.0f ms/step %.1f" % (d["value"], d["ms_per_step"]), d["pass_ms_per_step"], "rewalk", g("rewalk"), "presearch", g("presearch"), "scan", g("presearch_scan"), "classify", g("classify"), "cand", s["candidates"], "rewalked", s["rewalked"], "memo", s["memo_hits"], s["memo_adds"], "pre", s.get("pre_requests"), s.get("pre_adds"), "unitigs", d["config"]["unitigs"], d["config"]["unitig_bp"])
except Exception as e:
    print("presearch", sys.argv[2], "ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
for v in 1 0; do
ABG_PRESEARCH=$v ABG_WALK_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/dbg_ps$v.json 2> $O/dbg_ps$v.err
grep walkdbg $O/dbg_ps$v.err | grep -v "lookAhead" > $O/walkdbg_ps$v.txt
echo "== walkdbg presearch=$v"; grep -A1 "rewalk" $O/walkdbg_ps$v.txt | grep -v "^--" | cut -c1-260 | head -40; grep slowest $O/walkdbg_ps$v.txt | cut -c1-200
done
