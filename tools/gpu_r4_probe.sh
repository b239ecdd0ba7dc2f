#!/bin/bash
# Round-4 probe on a gpurun box: the new parity cases, then one short bench per knob (what each costs or gains on
# THIS box), the walkers' first-launch breakdown, and the FWalk write traffic of two builds.  -> gpurun_out/r4p/
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4p
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernel_ms"]
    g = lambda n: k.get(n, {}).get("ms", 0)
    print("%-28s %7.1f Mk/s  step %6.1f  pass1 %6.1f pass2 %6.1f | staged %5.1f purity %5.1f target %5.1f apply %5.1f retry %5.1f claim %5.1f | guide %5.1f rewalk %6.1f presearch %5.1f classify %6.1f | pending %d rounds %d cand %d parity %s" % (
        sys.argv[1].split("/")[-1][:28], d["value"], d["ms_per_step"], d["pass_ms_per_step"]["pass1"], d["pass_ms_per_step"]["pass2"],
        g("hash_bin_staged"), g("tile_purity"), g("op_target"), g("tile_apply"), g("insert_retry") + g("insert_round"), g("claim_list"),
        g("guide_build"), g("rewalk"), g("presearch") + g("presearch_scan"), g("classify"),
        d["engine_stats"]["tiled_pending"], d["engine_stats"]["insert_rounds"], d["engine_stats"]["candidates"], d.get("parity", {}).get("ok")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
run() { name=$1; shift; env "$@" timeout 300 $B > $O/$name.json 2> $O/$name.err; line $O/$name.json | tee -a $O/summary.txt; }
run base_benign1 ABG_BENIGN=1
run base_benign0 ABG_BENIGN=0
run guide8 ABG_GUIDE_STRIDE=8
run guide16 ABG_GUIDE_STRIDE=16
run pipeline2 ABG_PIPELINE=2
run first32k ABG_P2_FIRST_BATCH=32768
run w2 ABG_LIB=$R/abyss_amd/lib/variants/libabyss_amd_w2.so
[ -f $R/abyss_amd/lib/variants/libabyss_amd_inl.so ] && run inl ABG_LIB=$R/abyss_amd/lib/variants/libabyss_amd_inl.so
# the walkers' own breakdown, launch by launch (stderr)
ABG_WALK_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/walkdbg.json 2> $O/walkdbg.err; grep walkdbg $O/walkdbg.err | cut -c1-330 > $O/walkdbg.txt; head -40 $O/walkdbg.txt
# FETCH / WRITE of the walkers in the shipped build and in the all-inlined one (no call-site spills)
cd /tmp
for v in main inl; do
  lib=$R/abyss_amd/lib/libabyss_amd.so; [ $v = inl ] && lib=$R/abyss_amd/lib/variants/libabyss_amd_inl.so
  [ -f $lib ] || continue
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pmc_$v$c
    ABG_LIB=$lib timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$v$c -o pmc -- python $R/bench.py --warmup 0 --steps 1 --no-cpu-baseline --no-end-to-end --no-events > /tmp/pmc_$v$c.log 2>&1
  done
done
cd $R
python - <<'PY' | tee $O/pmc_walk_variants.txt
import csv, glob, collections
for v in ("main", "inl"):
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob('/tmp/pmc_%s%s/**/*counter_collection.csv' % (v, c), recursive=True):
            for row in csv.DictReader(open(f)):
                if row['Counter_Name'] != c: continue
                n = row['Kernel_Name']
                for key in ("FWalk", "FPresearch<", "FClassify", "FTileApply", "FTilePurity", "FOpTarget", "FBinCoarse", "FBinFine", "FHashOps", "FInsertRound", "FClaimList"):
                    if key in n: agg[key][0] += float(row['Counter_Value']); agg[key][1] += 1
        print(v, c, " ".join("%s=%.1fGB/%d" % (k, a[0] * 1024 / 1e9, a[1]) for k, a in sorted(agg.items())))
PY
