#!/bin/bash
# A quick look on a gpurun box (round 5): the parity tests that matter most, a default bench line, one more per
# "name:VAR=value[,VAR=value...]" argument, and the walkers' own breakdown (ABG_WALK_DEBUG).  -> gpurun_out/$OUT (default r5q)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r5q}
mkdir -p $O
export TMPDIR=/tmp
cd $R
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
fi
show() { python - $1 <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d["kernel_ms"]
print("%s: %.1f Mk/s step %.1f pass1 %.1f pass2 %.1f parity %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["pass_ms_per_step"]["pass1"], d["pass_ms_per_step"]["pass2"], d.get("parity", {}).get("ok")))
print("   " + " ".join("%s=%.1f/%d" % (n, v["ms"], v["launches"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"])[:22]))
es = d.get("engine_stats", {})
print("   " + " ".join("%s=%s" % (n, es.get(n)) for n in ("candidates", "walked", "rewalked", "generated", "bulk_steps", "lin_steps", "chain_steps", "memo_hits", "memo_adds", "pre_requests", "tiled_pending", "insert_rounds")))
PY
}
STEPS=${STEPS:-3}
timeout 300 python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_default.json 2> $O/bench_default.err; show $O/bench_default.json
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  env ${vars//,/ } timeout 300 python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_$name.json 2> $O/bench_$name.err; show $O/bench_$name.json
done
if [ "${SKIP_WALKDBG:-0}" != 1 ]; then
  ABG_WALK_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/walkdbg.json 2> $O/walkdbg.err; grep walkdbg $O/walkdbg.err | cut -c1-400 > $O/walkdbg.txt; head -24 $O/walkdbg.txt
fi
