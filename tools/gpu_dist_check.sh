#!/bin/bash
# Partitioned-run evidence on a 1-GPU gpurun box: the gpu tests of the partitioned path, the
# default 1-GPU bench line (unchanged path), the partitioned code path forced onto one rank with
# the library's RCCL communicator at full size (+ rocprofv3 kernel stats of that command), and a
# 2-rank run of bench.py with both ranks on the one GPU (gloo, host-staged collectives, small job).
# Summaries -> gpurun_out/dist/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dist
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 420 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -15 > $O/pytest_gpu_dist.log; cat $O/pytest_gpu_dist.log
ABG_FORCE_DIST=1 timeout 200 python bench.py --no-cpu-baseline > $O/bench_forced_partitioned_1rank.json 2> $O/bench_forced.err; cut -c1-400 $O/bench_forced_partitioned_1rank.json; tail -3 $O/bench_forced.err
ABG_BENCH_BACKEND=gloo timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --comm staged --pairs 400000 --bloom 256M --steps 1 --warmup 1 > $O/bench_2ranks_1gpu_staged.json 2> $O/bench_2ranks.err; cut -c1-700 $O/bench_2ranks_1gpu_staged.json; tail -3 $O/bench_2ranks.err
timeout 200 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json; tail -3 $O/bench_default.err
python - <<'PY'
import json
for f in ("bench_default", "bench_forced_partitioned_1rank"):
    try:
        d = json.load(open("gpurun_out/dist/%s.json" % f))
        print(f, "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 1), "unitigs", d["config"]["unitigs"], d["config"]["unitig_bp"])
        print("  ", {k: v["ms"] for k, v in d["kernel_ms"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp
ABG_FORCE_DIST=1 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o trace -- python $R/bench.py --no-cpu-baseline --warmup 0 > /tmp/prof_trace.log 2>&1
find /tmp/prof_trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_forced_partitioned.csv \;
head -14 $O/kernel_stats_forced_partitioned.csv | cut -c1-170
