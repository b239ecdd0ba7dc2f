"""Shared helpers of the test-suite: golden cases and comparisons."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class GoldenCase:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN, name + ".reads.npz"))
        self.buf = z["buf"].tobytes()
        self.off = z["off"].astype(np.uint64)
        self.n = len(self.off) - 1
        self.meta = json.load(open(os.path.join(GOLDEN, name + ".json")))
        self.fasta = open(os.path.join(GOLDEN, name + ".fa"), "rb").read()
        self.readlog = open(os.path.join(GOLDEN, name + ".readlog.tsv"), "rb").read()
        self.trace = open(os.path.join(GOLDEN, name + ".trace.tsv"), "rb").read()
        self.ids = [b"r%d" % i for i in range(self.n)]
        self.reads = [self.buf[int(self.off[i]):int(self.off[i + 1])] for i in range(self.n)]
        o = dict(k=0, num_hashes=4, min_cov=2, trim=None, bloom_bytes=0, mask=None)
        for opt in self.meta["options"]:
            if opt.startswith("-k") and not opt.startswith("--"):
                o["k"] = int(opt[2:])
            elif opt.startswith("-b"):
                o["bloom_bytes"] = int(opt[2:-1]) * (1 << 20)
            elif opt.startswith("-H"):
                o["num_hashes"] = int(opt[2:])
            elif opt.startswith("--kc="):
                o["min_cov"] = int(opt[5:])
            elif opt.startswith("-t"):
                o["trim"] = int(opt[2:])
            elif opt.startswith("-K"):
                o["K"] = int(opt[2:])
            elif opt.startswith("--qr-seed="):
                o["qr"] = int(opt[10:])
        self.opts = o

    def kwargs(self):
        o = self.opts
        return dict(k=o["k"], bloom_bytes=o["bloom_bytes"], num_hashes=o["num_hashes"], min_cov=o["min_cov"],
                    trim=o["trim"])


def mask_of(g):
    """Spaced seed of a golden case (SpacedSeed.h:18-75), or None."""
    from abyss_amd import api
    if "K" in g.opts:
        return api.spaced_seed_kmer_pair(g.opts["k"], g.opts["K"])
    if "qr" in g.opts:
        return api.spaced_seed_qr_pair(g.opts["k"], g.opts["qr"])
    return None


def contig_tuple(c):
    return (c.contig_id, c.read_index, bytes(c.seq), c.coverage, c.redundant, c.left_ext, c.right_ext,
            c.left_code, c.right_code, c.seed_pos)


def random_reads(n, L, seed, alphabet=b"ACGT"):
    rng = np.random.default_rng(seed)
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return a[rng.integers(0, len(a), size=(n, L))]
