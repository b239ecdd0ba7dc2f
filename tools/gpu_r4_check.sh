#!/bin/bash
# GPU suite + short bench lines (per-kernel split on one line each) + where the drop-in binary's wall time goes.  -> gpurun_out/r4c/
# usage: gpu_r4_check.sh [VAR=value ...]   (one extra bench per VAR=value, e.g. ABG_OVERLAP_PURITY=0)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
show() { python - $1 <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d["kernel_ms"]
print("%s: %.1f Mk/s step %.1f pass1 %.1f pass2 %.1f parity %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["pass_ms_per_step"]["pass1"], d["pass_ms_per_step"]["pass2"], d.get("parity", {}).get("ok")))
print("   " + " ".join("%s=%.1f/%d" % (n, v["ms"], v["launches"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"])[:22]))
PY
}
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_default.json 2> $O/bench_default.err; show $O/bench_default.json
for kv in "$@"; do
  env $kv timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_${kv%%=*}.json 2> $O/bench_${kv%%=*}.err; show $O/bench_${kv%%=*}.json
done
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-events > $O/bench_noevents.json 2> $O/bench_noevents.err; show $O/bench_noevents.json
bash tools/gpu_e2e_timing.sh 2>&1 | grep -v "^\[host.*packed\|^$" | cut -c1-200 | tee $O/e2e.log | tail -40
