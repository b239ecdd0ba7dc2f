#!/bin/bash
# Evidence at the head of round 4: the whole GPU suite, the default bench line as the driver runs it, a no-events line, the
# rocprofv3 kernel statistics of two steps.  -> gpurun_out/r4f/
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-events > $O/bench_noevents.json 2> $O/bench_noevents.err
python - <<'PY'
import json
for f in ("bench_default", "bench_noevents"):
    d = json.loads([l for l in open("gpurun_out/r4f/%s.json" % f) if l.startswith("{")][-1])
    print(f, round(d["value"], 1), round(d["ms_per_step"], 1), d.get("pass_ms_per_step"), d.get("parity", {}).get("ok"), d["roofline"]["kernel"], d["roofline"]["frac"], (d.get("end_to_end") or {}).get("wall_ms"))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r4f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/prof_bench.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -8 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/prof
