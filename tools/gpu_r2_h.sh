#!/bin/bash
set -u
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
show() { python - $1 $2 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]
print(sys.argv[2], "ms/step %.0f" % d["ms_per_step"], d["pass_ms_per_step"], "setup", d["setup_ms_per_step"], "kernel sum %.0f" % sum(b["ms"] for a,b in k.items() if a!="classify"))
PY
}
timeout 600 python bench.py --no-cpu-baseline --steps 2 > $O/bench.json 2> $O/bench.err; show $O/bench.json events
timeout 600 python bench.py --no-cpu-baseline --steps 2 --no-events > $O/bench_ne.json 2> $O/bench_ne.err; show $O/bench_ne.json noevents
