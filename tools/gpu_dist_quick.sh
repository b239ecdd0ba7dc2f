#!/bin/bash
# Short follow-up on a 1-GPU box: the partitioned code path forced onto one rank with the library's
# RCCL communicator at full size (bench line), then the single-process gpu tests of that path.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dist
mkdir -p $O
export TMPDIR=/tmp
cd $R
ABG_FORCE_DIST=1 timeout 150 python bench.py --no-cpu-baseline > $O/bench_forced_partitioned_1rank.json 2> $O/bench_forced.err; cut -c1-300 $O/bench_forced_partitioned_1rank.json; tail -3 $O/bench_forced.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/dist/bench_forced_partitioned_1rank.json"))
    print("forced value", round(d["value"], 1), "ms", round(d["ms_per_step"], 1), "unitigs", d["config"]["unitigs"], d["config"]["unitig_bp"])
    print("  ", {k: (v["ms"], v["launches"]) for k, v in d["kernel_ms"].items()})
except Exception as e:
    print("unreadable", e)
PY
timeout 120 python -m pytest tests/test_gpu_dist.py -x -q -k "k64 or share_reads" 2>&1 | tail -8 > $O/pytest_gpu_dist_quick.log; cat $O/pytest_gpu_dist_quick.log
