#!/bin/bash
# GPU suite + a short bench line (per-kernel split on one line).  -> gpurun_out/r4c/
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end "$@" > $O/bench$i.json 2> $O/bench$i.err
python - $O/bench$i.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d["kernel_ms"]
print("%.1f Mk/s step %.1f pass1 %.1f pass2 %.1f parity %s" % (d["value"], d["ms_per_step"], d["pass_ms_per_step"]["pass1"], d["pass_ms_per_step"]["pass2"], d.get("parity", {}).get("ok")))
print(" ".join("%s=%.1f/%d" % (n, v["ms"], v["launches"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"])[:24]))
print(d["engine_stats"])
PY
done
