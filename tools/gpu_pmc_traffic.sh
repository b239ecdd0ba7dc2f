#!/bin/bash
# HBM traffic of the PASS 1 / classify kernels from the TCC counters, each in its own pass
# (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2).  Summaries go to gpurun_out/prof/.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---warmup 0 --steps 1 --no-cpu-baseline --no-end-to-end}"
export PMC_ARGS="$ARGS"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 1500 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py $ARGS > /tmp/pmc_$c.log 2>&1
  tail -1 /tmp/pmc_$c.log | cut -c1-200
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != c: continue
            name = row['Kernel_Name']
            for key in ("FHashOps", "FBinCoarse", "FBinFine", "FTilePurity", "FOpTarget", "FTileApply", "FClaimList", "FHashClaim", "FInsertRound", "FClassify", "FWalk", "FGuideBuild", "k_commit", "k_insert_drain", "FContigPrep",
                        "FPcTimeMin", "FPcDecide", "FPcApply", "FCoSettle", "FCoFinal", "FPendCount", "FPendWrite", "FPreCommit", "FPcShort", "FRefilter", "FSolidPlane"):
                if key in name:
                    agg[key][0] += float(row['Counter_Value']); agg[key][1] += 1
        for k, (v, n) in agg.items():
            out.setdefault(k, {})[c] = {"sum": v, "dispatches": n}
# what the numbers are tied to: the kernel sources they were taken on (bench.py's pick_evidence compares the digest) and, when the
# caller passed it (COMMIT=$(git rev-parse --short HEAD) in the gpurun command: the box has no .git), the commit
import os, sys
sys.path.insert(0, '.')
import bench
out["_csrc_sha256"] = bench.csrc_digest()
if os.environ.get("COMMIT"): out["_commit"] = os.environ["COMMIT"]
out["_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units) over: python bench.py " + os.environ.get("PMC_ARGS", "")
json.dump(out, open('gpurun_out/prof/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
