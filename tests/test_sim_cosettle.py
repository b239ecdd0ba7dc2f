"""tools/sim_cosettle.c: PASS 1's round-5 rule (op_verdict, abg_engine.h) restated in ~100 lines of C with its own
hashes and run against the sequential conservative update (CountingBloomFilter.hpp:135-162), batch by batch -- the check the rule went
through before any device code existed.  Here at sizes that take a second: a sparse filter, configs[2]'s occupancy, a dense one (chains of
k-mers sharing counters: the fixed point needs many passes) and counters driven to 255."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sim") / "sim")
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "sim_cosettle.c")], check=True)
    return exe


@pytest.mark.parametrize("genome,batches,bytes_per_base,coverage", [(60000, 20, 71.6, 50), (60000, 20, 35.8, 50), (30000, 12, 8, 50), (6000, 10, 71.6, 700)])
def test_settling_kmers_that_raise_shared_counters_gives_the_sequential_filter(sim, genome, batches, bytes_per_base, coverage):
    r = subprocess.run([sim, str(genome), str(batches), str(bytes_per_base), "1", str(coverage)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "counters equal to sequential" in out, (out, r.stderr.decode())
    old, new = (float(x) for x in re.search(r"pending old rule ([\d.]+)%.*pending after closure ([\d.]+)%", out).groups())
    assert new <= old, out
    if bytes_per_base > 30 and coverage <= 50:
        assert new * 5 < old, out  # (the regime of BASELINE's configurations: 6.8 % -> 0.2 %, 12.8 % -> 0.8 %)


def test_the_simulation_with_the_rule_off_is_the_round_4_state(sim):
    r = subprocess.run([sim, "30000", "10", "35.8", "0", "50"], stdout=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "candidates by the new rule 0.00%" in out and "counters equal to sequential" in out, out
