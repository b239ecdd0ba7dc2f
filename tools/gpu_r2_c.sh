#!/bin/bash
set -u
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
show() { python - $1 $2 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
print(sys.argv[2], "ms/step %.0f" % d["ms_per_step"], "rewalk %.0f/%d" % (k["rewalk"]["ms"], k["rewalk"]["launches"]), "classify %.0f" % k["classify"]["ms"], "cand", s["candidates"], "memo", s["memo_hits"], s["memo_adds"], "unitigs", d["config"]["unitigs"], d["config"]["unitig_bp"])
print({a:round(b["ms"]) for a,b in k.items() if b["ms"]>3})
PY
}
timeout 600 python bench.py --no-cpu-baseline --steps 1 > $O/bench.json 2> $O/bench.err; show $O/bench.json memo
ABG_MEMO=0 timeout 600 python bench.py --no-cpu-baseline --steps 1 > $O/bench_nomemo.json 2> $O/bench_nomemo.err; show $O/bench_nomemo.json nomemo
ABG_WALK_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/dbg.json 2> $O/dbg.err; grep walkdbg $O/dbg.err | cut -c1-460 | head -30
