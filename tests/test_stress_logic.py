"""Randomised parameter sweep of the device logic (serial host execution) against the oracle:
odd and even k, saturated filters (heavy false-positive branching), trim 0..2k, 1..9 hash
functions, tiny claim tables and insert batches."""
import numpy as np
import pytest

import oracle_binding as ob
from abyss_amd import api, synth
from test_hostcheck import HostCheck
from util import contig_tuple


@pytest.mark.parametrize("trial", range(10))
def test_random_configuration(trial):
    rng = np.random.default_rng(1000 + trial)
    G = int(rng.integers(3000, 9000))
    k = int(rng.choice([9, 12, 15, 16, 21, 24, 31, 32, 33, 40, 48, 63, 64, 65, 80]))
    cov = float(rng.choice([15, 25]))
    err = float(rng.choice([0.005, 0.02, 0.05]))
    kc = int(rng.choice([1, 2, 3]))
    trim = int(rng.choice([0, 1, 3, k // 2, k, 2 * k]))
    H = int(rng.choice([1, 2, 4, 5, 9]))
    h1, h2 = synth.make_genome(G, seed=trial, snp_every=int(rng.choice([0, 100, 500])))
    m1, m2 = synth.sample_pairs(h1, h2, int(G * cov / 200), read_len=100, err=err, seed=trial + 100, frag_lo=150, frag_hi=250)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    counters = int(rng.choice([1 << 17, 1 << 19, 99991 * 8]))
    o = ob.Oracle(k, counters=counters, num_hashes=H, min_cov=kc, trim=trim)
    hc = HostCheck(k, counters, H, kc, trim, insert_batch=int(rng.choice([3000, 50000])), claim_log2=int(rng.choice([8, 14])),
                   p2_first=int(rng.choice([16, 256, 100000])))
    o.load(buf, off)
    hc.load(buf, off)
    assert np.array_equal(o.counters(), hc.counters())
    ro, co = o.assemble(buf, off)
    rh, ch = hc.assemble(buf, off)
    assert np.array_equal(ro, rh)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch]
    assert np.array_equal(o.visited(), hc.visited())
    assert o.assembly_counters() == hc.assembly_counters()
