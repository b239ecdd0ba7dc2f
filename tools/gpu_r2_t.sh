#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2t; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2t/bench.json"))
print("ms/step %.1f" % d["ms_per_step"], d["pass_ms_per_step"], d["value"])
PY
