#!/bin/bash
# configs[2] (200 M pairs, k=64, B=40G) at stated size on one GPU, twice: the plain run and the partitioned
# code path on one rank (ABG_FORCE_DIST=1, every collective an identity).  No reference run exists at this
# size (days of CPU): the two runs must agree on the digests of --invariants.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c2inv; mkdir -p $O
cd $R
timeout 900 python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --invariants > $O/plain.json 2> $O/plain.err
cut -c1-300 $O/plain.json
ABG_FORCE_DIST=1 timeout 1200 python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --invariants > $O/forced.json 2> $O/forced.err
cut -c1-300 $O/forced.json
python - $O/plain.json $O/forced.json <<'PY'
import json, sys
a, b = (json.load(open(p)) for p in sys.argv[1:3])
print("plain  ", a["value"], a["ms_per_step"], a["invariants"])
print("forced ", b["value"], b["ms_per_step"], b["invariants"])
print("AGREE" if a["invariants"] == b["invariants"] else "DIFFER")
PY
tail -n 3 $O/plain.err; tail -n 3 $O/forced.err
