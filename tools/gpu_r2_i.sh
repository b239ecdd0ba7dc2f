#!/bin/bash
# guide/bulk steps under spaced seeds and odd k: parity, then configs[3] and the default bench
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
ABG_PRINT_STATS=1 timeout 600 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err; cut -c1-900 $O/bench3.json; echo
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err; cut -c1-900 $O/bench1.json; echo
