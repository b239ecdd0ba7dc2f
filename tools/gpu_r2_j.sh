#!/bin/bash
# the partitioned run's tiles: GPU dist tests, then the partitioned code path on one rank next to the plain run
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
ABG_FORCE_DIST=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_forced.json 2> $O/bench_forced.err; cut -c1-400 $O/bench_forced.json; echo
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2j/bench_forced.json"))
print(d["ms_per_step"], d.get("pass_ms_per_step"), d["engine_stats"])
print({k:(round(v["ms"],1),v["launches"]) for k,v in d["kernel_ms"].items() if v["ms"]>1})
PY
