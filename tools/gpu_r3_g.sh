#!/bin/bash
# error returns, full-size parity against the reference's -j1 digest, and the default bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_errors.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15 | cut -c1-300 > $O/pytest.log; cat $O/pytest.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value %.0f Mk-mers/s, %.1f ms/step; no_events %s; parity %s" % (d["value"], d["ms_per_step"], d.get("no_events"), d.get("parity")))
print("cpu_baseline", {k:(v if not isinstance(v,dict) else '{..}') for k,v in d["cpu_baseline"].items()})
print("end_to_end", d.get("end_to_end"))
print("roofline", {k:v for k,v in d["roofline"].items() if k!="kernels"})
PY
