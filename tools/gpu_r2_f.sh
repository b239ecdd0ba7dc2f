#!/bin/bash
set -u
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -2
timeout 1200 python bench.py --config 2 --no-cpu-baseline --steps 1 --warmup 0 > $O/bench_config2.json 2> $O/bench_config2.err; cut -c1-1200 $O/bench_config2.json; tail -3 $O/bench_config2.err
