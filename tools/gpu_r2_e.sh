#!/bin/bash
set -u
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q 2>&1 | tail -15 > $O/pytest_scale.log; cat $O/pytest_scale.log
timeout 900 python bench.py --config 3 --no-cpu-baseline --steps 1 > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-900 $O/bench_config3.json; tail -2 $O/bench_config3.err
