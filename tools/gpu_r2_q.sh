#!/bin/bash
# PASS 1 with the next batch staged on the side stream: parity, then the bench with and without
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2q; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
for v in 1 0; do
ABG_OVERLAP_BINS=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_overlap$v.json 2> $O/bench_overlap$v.err
python - $O/bench_overlap$v.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_ms"]
print("overlap", sys.argv[2], "ms/step %.1f" % d["ms_per_step"], d["pass_ms_per_step"], {n:round(v["ms"],1) for n,v in k.items() if n in ("hash_ops","bin_coarse","bin_fine","hash_bin_staged","tile_purity","op_target","tile_apply","insert_retry")})
PY
done
ABG_OVERLAP_BINS=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events > $O/bench_noevents.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_noevents.json')); print('no events: ms/step %.1f' % d['ms_per_step'], d.get('pass_ms_per_step'))"
