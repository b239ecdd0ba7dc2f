#!/bin/bash
set -u
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
show() { python - $1 $2 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
print(sys.argv[2], "ms/step %.0f" % d["ms_per_step"], "rewalk %.0f/%d" % (k["rewalk"]["ms"], k["rewalk"]["launches"]), "classify %.0f" % k["classify"]["ms"], "cand", s["candidates"], "rewalked", s["rewalked"], "unitigs", d["config"]["unitigs"], d["config"]["unitig_bp"])
PY
}
for depth in 2 1 3; do
ABG_PIPELINE=$depth timeout 600 python bench.py --no-cpu-baseline --steps 1 > $O/bench_p$depth.json 2> $O/bench_p$depth.err; show $O/bench_p$depth.json depth$depth
done
