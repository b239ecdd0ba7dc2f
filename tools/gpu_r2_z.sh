#!/bin/bash
# PASS 2 with the next batch's walkers queued beside this batch's slowest ones (ABG_TAIL_FILL=1): parity, then the bench both ways
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2z; mkdir -p $O
cd $R
ABG_TAIL_FILL=1 timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "reproduces or scale or oracle" 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
for v in 1 0; do
ABG_TAIL_FILL=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_tf$v.json 2> $O/bench_tf$v.err
python - $O/bench_tf$v.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernel_ms"]; s=d["engine_stats"]
    print("tail_fill", sys.argv[2], "ms/step %.1f" % d["ms_per_step"], d["pass_ms_per_step"], "rewalk %.0f/%d" % (k["rewalk"]["ms"], k["rewalk"]["launches"]), "classify %.0f" % k["classify"]["ms"], "cand", s["candidates"], "rewalked", s["rewalked"], "breaks", s["commit_breaks"], "unitigs", d["config"]["unitigs"])
except Exception as e:
    print("tail_fill", sys.argv[2], "ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-500:])
PY
done
