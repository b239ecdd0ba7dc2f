"""The counter-based read generator (abyss_amd/synth.py): its numpy form (FASTQ files for the
reference binary) and its torch form (packed reads generated where bench.py times them) produce the
same reads, slices of a read set equal the whole, and its statistics are those of the recipe
(SURVEY.md section 8d)."""
import numpy as np
import torch

from abyss_amd import synth


def _unpack(words, n, read_len):
    wpr = (read_len + 15) // 16
    w = words.cpu().numpy().view(np.uint32).reshape(n, wpr)
    codes = (w[:, :, None] >> (2 * np.arange(16, dtype=np.uint32))[None, None, :]) & 3
    return codes.reshape(n, wpr * 16)[:, :read_len].astype(np.uint8)


def test_numpy_and_torch_generate_the_same_reads():
    h1, h2 = synth.make_genome(50000, seed=42)
    n, L = 3000, 150
    a, b = synth.sample_pairs_cb(h1, h2, n, read_len=L, err=0.005, seed=7)
    words, woff, lens = synth.packed_reads_torch(h1, h2, n, L, 0.005, 7, torch.device("cpu"), chunk=1024)
    got = _unpack(words, 2 * n, L)
    assert np.array_equal(got[:n], a) and np.array_equal(got[n:], b)
    assert int(woff[-1]) == 2 * n * 10 and int(lens[0]) == L
    # a slice of the set (what a rank of a partitioned bench generates) is the same reads
    a2, b2 = synth.sample_pairs_cb(h1, h2, 1000, read_len=L, err=0.005, seed=7, first=1500, total_pairs=n)
    assert np.array_equal(a2, a[1500:2500]) and np.array_equal(b2, b[1500:2500])
    w2, _, _ = synth.packed_reads_torch(h1, h2, 1000, L, 0.005, 7, torch.device("cpu"), first=1500, total_pairs=n)
    got2 = _unpack(w2, 2000, L)
    assert np.array_equal(got2[:1000], a[1500:2500]) and np.array_equal(got2[1000:], b[1500:2500])


def test_statistics_follow_the_recipe():
    h1, h2 = synth.make_genome(200000, seed=42)
    frag, start, hap, strand = synth.pair_draws(200000, 200000, 7)
    assert frag.min() == 350 and frag.max() == 450 and start.min() >= 0 and start.max() < 200000 - 450
    assert abs(hap.mean() - 0.5) < 0.01 and abs(strand.mean() - 0.5) < 0.01
    e, off = synth.error_draws(0, 20000, 150, 0.005, 7)
    assert abs(e.mean() - 0.005) < 0.0003 and set(np.unique(off)) == {1, 2, 3}
    # mates: read i and read n + i come from the two ends of one fragment, on opposite strands
    a, b = synth.sample_pairs_cb(h1, h2, 2000, err=0.0, seed=7)
    frag, start, hap, strand = synth.pair_draws(2000, 200000, 7)
    for i in (0, 17, 1999):
        g = h2 if hap[i] else h1
        left = g[start[i]:start[i] + 150]
        right = (3 - g[start[i] + frag[i] - 150:start[i] + frag[i]])[::-1]
        m1, m2 = (right, left) if strand[i] else (left, right)
        assert np.array_equal(a[i], m1) and np.array_equal(b[i], m2)
