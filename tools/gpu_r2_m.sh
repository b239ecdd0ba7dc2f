#!/bin/bash
# configs[2] at its stated size after the memory fixes (allocations logged), and the scale test
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2m; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
ABG_MEM_DEBUG=1 timeout 1500 python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_config2.json 2> $O/bench_config2.err; cut -c1-900 $O/bench_config2.json; echo; grep -v "^\[mem\]" $O/bench_config2.err | tail -5; grep -c "^\[mem\]" $O/bench_config2.err; grep "^\[mem\]" $O/bench_config2.err | sort -t+ -k2 -n -r | head -12
