#!/bin/bash
# First-contact script for a gpurun box: tests, smoke, a reduced bench and a rocprofv3 trace.
# Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== host"; nproc; free -g | head -2; rocminfo | grep -m2 -E "gfx|Marketing" 
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== bench (reduced: 500k pairs, B=256M)"
timeout 900 python bench.py --pairs 500000 --bloom 256M --warmup 0 --steps 1 --no-cpu-baseline 2>&1 | tail -3
} > gpurun_out/check.log 2>&1
cat gpurun_out/check.log | tail -60
