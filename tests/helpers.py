"""Shared test helpers: 2-bit packing (A=0 C=1 G=2 T=3, first base in the LSBs,
as in the reference's UnitigSet), fixture readers, seeded synthetic unitig sets and query mixes."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE[ord(_c)] = _i
    _CODE[ord(_c.lower())] = _i


def encode_kmer(s):
    """forward word of a k-mer string (first base in the LSBs)."""
    w = 0
    for i, c in enumerate(s):
        code = int(_CODE[ord(c)])
        assert code < 4, s
        w |= code << (2 * i)
    return w


def decode_kmer(w, k):
    return "".join("ACGT"[(w >> (2 * i)) & 3] for i in range(k))


def rc_word(w, k):
    r = 0
    for i in range(k):
        r |= (3 - ((w >> (2 * i)) & 3)) << (2 * (k - 1 - i))
    return r


def pack_codes(codes):
    """uint8 codes (0..3) -> uint64 words, padded with 2 zero words."""
    n = len(codes)
    nw = (n + 31) // 32
    padded = np.zeros(nw * 32, dtype=np.uint64)
    padded[:n] = codes
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))
    words = (padded.reshape(nw, 32) << shifts).sum(axis=1, dtype=np.uint64)  # disjoint bits: sum == or
    return np.concatenate([words, np.zeros(2, dtype=np.uint64)])


def pack_seqs(seqs):
    """list of ACGT strings -> (words, total_len, starts[n+1])."""
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    starts = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=starts[1:])
    raw = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    codes = _CODE[raw]
    assert (codes < 4).all(), "non-ACGT base in unitig"
    return pack_codes(codes), int(starts[-1]), starts


def load_cf_seqs(prefix):
    """cuttlefish reduced-GFA segments `id\\tSEQ` (unitig_set.rs:119-165) -> (k, [seq...])."""
    with open(prefix + ".json") as f:
        k = json.load(f)["parameters info"]["k"]
    seqs = []
    with open(prefix + ".cf_seg") as f:
        for line in f:
            line = line.rstrip("\n")
            if line:
                seqs.append(line.split("\t", 1)[1])
    return k, seqs


def read_fasta(path):
    """FastaReader (util.rs:88-149): header verbatim, sequence lines concatenated."""
    recs, name, buf = [], None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(buf)))
                name, buf = line[1:], []
            elif name is not None:
                buf.append(line)
    if name is not None:
        recs.append((name, "".join(buf)))
    return recs


def random_unitigs(seed, n_unitigs, mean_len, k, max_len=5000):
    """seeded random ACGT unitig codes; returns (codes uint8, starts)."""
    rng = np.random.default_rng(seed)
    lens = k + np.minimum(rng.exponential(mean_len, n_unitigs).astype(np.int64), max_len)
    starts = np.zeros(n_unitigs + 1, dtype=np.uint64)
    np.cumsum(lens, out=starts[1:].view(np.int64))
    codes = rng.integers(0, 4, int(starts[-1]), dtype=np.uint8)
    return codes, starts


def kmer_words_at(words, positions, k):
    """forward words of the k-mers starting at global positions (numpy, vectorised)."""
    positions = np.asarray(positions, dtype=np.uint64)
    b = positions * np.uint64(2)
    wi = (b >> np.uint64(6)).astype(np.int64)
    sh = b & np.uint64(63)
    lo = words[wi] >> sh
    hi = np.where(sh > 0, words[wi + 1] << ((np.uint64(64) - sh) & np.uint64(63)), np.uint64(0))
    return (lo | hi) & np.uint64((1 << (2 * k)) - 1)


def rc_words(x, k):
    """vectorised reverse complement of forward words."""
    x = ~np.asarray(x, dtype=np.uint64)
    m = np.uint64
    x = ((x >> m(2)) & m(0x3333333333333333)) | ((x & m(0x3333333333333333)) << m(2))
    x = ((x >> m(4)) & m(0x0F0F0F0F0F0F0F0F)) | ((x & m(0x0F0F0F0F0F0F0F0F)) << m(4))
    x = x.byteswap()
    return x >> m(64 - 2 * k)


def valid_kmer_positions(starts, k):
    """all global positions whose k-mer lies inside one unitig."""
    starts = np.asarray(starts, dtype=np.int64)
    out = [np.arange(s, e - k + 1) for s, e in zip(starts[:-1], starts[1:]) if e - s >= k]
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


def query_mix(words, starts, k, n, seed, neg_frac=0.5):
    """n forward words: positives drawn from valid positions (random orientation),
    negatives uniform random 2k-bit words (NEG-U); shuffled."""
    rng = np.random.default_rng(seed)
    vp = valid_kmer_positions(starts, k)
    n_neg = int(n * neg_frac)
    n_pos = n - n_neg
    pos = vp[rng.integers(0, len(vp), n_pos)]
    fw = kmer_words_at(words, pos, k)
    flip = rng.integers(0, 2, n_pos).astype(bool)
    fw = np.where(flip, rc_words(fw, k), fw)
    neg = rng.integers(0, 1 << (2 * k), n_neg, dtype=np.uint64)
    q = np.concatenate([fw, neg])
    rng.shuffle(q)
    return q
