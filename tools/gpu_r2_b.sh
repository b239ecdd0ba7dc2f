#!/bin/bash
# walkers' time breakdown (ABG_WALK_DEBUG); args: extra env assignments are taken from the caller's environment
set -u
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
ABG_WALK_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/dbg_guide.json 2> $O/dbg_guide.err; grep walkdbg $O/dbg_guide.err | cut -c1-460 | head -${LINES_OUT:-12}
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2b/dbg_guide.json"))
print(d["value"], d["ms_per_step"], d["engine_stats"])
print({k:v["ms"] for k,v in d["kernel_ms"].items() if v["ms"]>3})
PY
