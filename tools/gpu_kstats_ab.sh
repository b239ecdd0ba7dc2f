#!/bin/bash
# rocprofv3 kernel statistics of one bench step per "name:VAR=value[,VAR=value...]" argument (and of the defaults first):
# the PASS-1 kernels side by side.  -> gpurun_out/$OUT (default kstats)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-kstats}
mkdir -p $O
export TMPDIR=/tmp
run() {
  name=$1; shift
  cd /tmp && env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o ks -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-events > $O/$name.log 2>&1
  f=$(find /tmp/ks_$name -name '*kernel_stats.csv' | head -1)
  t=$(find /tmp/ks_$name -name '*kernel_trace.csv' | head -1)
  if [ -n "${TRACE_OF:-}" ]; then python - "$t" "$TRACE_OF" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in sys.argv[2].split(","))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) // 2
print("dispatches of", sys.argv[2], "from the middle of the run: name, us")
for r in rows[mid:mid + 40]:
    print("   %-40s %8.1f" % (r["Kernel_Name"].split("abg::")[-1][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
  fi
  cp "$f" $O/${name}_kernel_stats.csv
  python - "$f" $name <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("==", sys.argv[2])
for r in rows[:26]:
    n = re.sub(r"void \(anonymous namespace\)::|abg::", "", r["Name"])[:64]
    print("%-64s %6s %9.1f ms  avg %9.1f us" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
}
run default A=1
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  run $name ${vars//,/ }
done
