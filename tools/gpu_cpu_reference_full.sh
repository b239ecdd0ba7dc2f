#!/bin/bash
# The unmodified reference (oracle/_ref/abyss-bloom-dbg) at -j<all cores> on the FULL configs[1] FASTQ files of the read set bench.py
# times (synth.make_read_set_cb), on the GPU box's host.  ~4 minutes.  -> gpurun_out/cpuref/cpu_reference_config1.json
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cpuref; mkdir -p $O
mkdir -p /tmp/cpuref && cd /tmp/cpuref
python - <<PY
import sys, time, json, os, subprocess, hashlib
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/oracle")
from abyss_amd import synth
import oracle_binding as ob
t = time.time()
h1, h2 = synth.make_genome(30_000_000, seed=42)
m1, m2 = synth.sample_pairs_cb(h1, h2, 5_000_000, read_len=150, err=0.005, seed=7)
synth.write_fastq("r1.fq", m1, "r", 1); synth.write_fastq("r2.fq", m2, "r", 2)
prep = time.time() - t
cores = os.cpu_count()
model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
t = time.time()
out, err = ob.run_ref(["-k64", "-b2G", "-H4", "-q3", "-v", "r1.fq", "r2.fq"], cwd=".", threads=cores)
wall = time.time() - t
kmers = 2 * 5_000_000 * (150 - 64 + 1)
lens = [int(l.split()[1]) for l in out.split(b"\n") if l.startswith(b">")]
res = {"what": "oracle/_ref/abyss-bloom-dbg (unmodified reference 2.3.10) -j%d on the two FASTQ files of configs[1]'s read set (synth.make_read_set_cb: the read set bench.py times), page cache warm, whole-binary wall time" % cores,
       "value": kmers / wall / 1e6, "unit": "Mk-mers/s", "wall_s": wall, "cores": cores, "cpu": model, "kind": "reference",
       "unitigs": len(lens), "unitig_bp": sum(lens), "fasta_sha256": hashlib.sha256(out).hexdigest(),
       "note": "-j>1 is not deterministic in the reference (the unitig set depends on the threads' interleaving); the -j1 digest is in tests/golden/full_size.json",
       "files_written_in_s": round(prep, 1)}
json.dump(res, open("$O/cpu_reference_config1.json", "w"), indent=1)
print(json.dumps(res))
PY
