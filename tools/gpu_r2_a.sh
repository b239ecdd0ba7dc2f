#!/bin/bash
# round 2, first contact: gpu tests, the default bench line, the walkers' debug counters
set -u
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 2 > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
ABG_WALK_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/bench_dbg.json 2> $O/bench_dbg.err; grep walkdbg $O/bench_dbg.err | cut -c1-400 | tail -24
ABG_GUIDE_STRIDE=0 timeout 600 python bench.py --no-cpu-baseline --steps 1 > $O/bench_noguide.json 2> $O/bench_noguide.err; cut -c1-400 $O/bench_noguide.json
