#!/bin/bash
# usage: tools/gpu_bench.sh <tag> <bench args...>   -- bench + rocprofv3 kernel trace of the same command
set -u
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -c 3000 gpurun_out/bench_$tag.json; tail -5 gpurun_out/bench_$tag.err
if [ "${PROFILE:-0}" = "1" ]; then
  cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o prof -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
  cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_$tag | head; find gpurun_out/prof_$tag -name "*stats*" | head
fi
