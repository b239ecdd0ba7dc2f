#!/bin/bash
set -u
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
show() { python - $1 $2 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
print(sys.argv[2], "ms/step %.0f" % d["ms_per_step"], "unitigs", d["config"]["unitigs"], d["config"]["unitig_bp"], {a:s[a] for a in ("tiled_ops","tiled_pending","tile_overflows","insert_rounds")})
print({a:(round(b["ms"]), b["launches"]) for a,b in k.items() if b["ms"]>3})
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --steps 1 > $O/bench.json 2> $O/bench.err; show $O/bench.json tiled; tail -2 $O/bench.err
