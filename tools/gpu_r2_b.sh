#!/bin/bash
# walkers' time breakdown (ABG_WALK_DEBUG) with and without the guide
set -u
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
ABG_WALK_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/dbg_guide.json 2> $O/dbg_guide.err; grep walkdbg $O/dbg_guide.err | cut -c1-420 | head -12
ABG_WALK_DEBUG=1 ABG_GUIDE_STRIDE=0 timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/dbg_noguide.json 2> $O/dbg_noguide.err; grep walkdbg $O/dbg_noguide.err | cut -c1-420 | head -12
