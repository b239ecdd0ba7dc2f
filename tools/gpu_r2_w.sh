#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2w; mkdir -p $O
cd $R
for v in 8 16 32 8; do
if [ $v = 8 ]; then unset ABG_LIB; else export ABG_LIB=$R/abyss_amd/lib/libabyss_amd_hc$v.so; fi
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_hc$v.json 2> $O/bench_hc$v.err
python - $O/bench_hc$v.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]
print("HC_RUN", sys.argv[2], "ms/step %.1f" % d["ms_per_step"], d["pass_ms_per_step"], {n:round(v["ms"],1) for n,v in k.items() if n in ("hash_bin_staged","insert_retry")}, d["config"]["unitigs"])
PY
done
