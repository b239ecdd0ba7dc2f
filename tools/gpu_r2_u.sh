#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2u; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
ABG_FORCE_DIST=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_forced.json 2> $O/bench_forced.err
python - <<'PY'
import json
for f in ("bench","bench_forced"):
    d=json.load(open("gpurun_out/r2u/%s.json" % f))
    print(f, "ms/step %.1f" % d["ms_per_step"], d["pass_ms_per_step"], round(d["value"],1))
PY
