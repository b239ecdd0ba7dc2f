#!/bin/bash
# A/B in configs[2]'s regime (35.8 filter bytes per genome base) at 1/40 of its size: this round's library against
# another build (ABG_LIB), per-kernel times.  usage (on the GPU box): bash tools/gpu_r4_regime_ab.sh [other.so]
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/regime; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --pairs 5000000 --bloom 1G --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d.get("kernel_ms", {})
print(sys.argv[2], "%.1f Mk/s step %.1f" % (d["value"], d["ms_per_step"]), d.get("pass_ms_per_step"))
print("  " + " ".join("%s=%.1f" % (n, v["ms"] / d["steps"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"])[:14]))
PY
}
run head
[ -n "$1" ] && run other ABG_LIB=$PWD/$1
run head_nodups ABG_LINK_DUPS=0
run head_again
