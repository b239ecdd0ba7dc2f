#!/bin/bash
# spaced seeds with rolled masked hashes: parity, configs[3]; then configs[2] at its stated size (one step, no warm-up)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2k; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-300 $O/bench_config3.json; echo
timeout 1500 python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_config2.json 2> $O/bench_config2.err; cut -c1-700 $O/bench_config2.json; echo; tail -3 $O/bench_config2.err
