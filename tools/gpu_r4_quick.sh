#!/bin/bash
# A quick look on a gpurun box: the parity tests that matter most, a default bench line, one more per VAR=value, and the
# walkers' own breakdown (ABG_WALK_DEBUG).  -> gpurun_out/r4q/
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4q
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
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
ABG_WALK_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/walkdbg.json 2> $O/walkdbg.err; grep walkdbg $O/walkdbg.err | cut -c1-400 > $O/walkdbg.txt; head -24 $O/walkdbg.txt
