#!/bin/bash
# end-of-round evidence: gpu tests, smoke, the default bench line (as the driver runs it), e2e, profiles
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-700 $O/bench.json; echo
bash tools/gpu_r2_e2e.sh 5000000 noref 2>&1 | grep -E "amd_|written" 
bash tools/gpu_r2_prof.sh final all 2>&1 | grep -E "k_walkers|FTileApply|FClassify|busy|GB$" | cut -c1-160
ABG_PRINT_STATS=1 timeout 600 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-300 $O/bench_config3.json; echo
