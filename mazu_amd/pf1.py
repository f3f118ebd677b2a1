"""Minimal reader for pufferfish (C++) dense-index directories -- plumbing for BASELINE config 0
and the README drop-in example.  Formats follow reference src/pf1/cpp.rs:190-237 (compact::vector),
:138-172 (cereal vectors), src/pf1/unitig_table.rs:28-49, src/pf1/mod.rs:213-234.
Host-side I/O only (numpy); nothing here is on the query path."""
import json
import os

import numpy as np


def read_compact(path):
    """compact::vector: u64 static_flag, width, len, capacity, then words; element i at bit i*width."""
    raw = np.fromfile(path, dtype=np.uint64)
    width, n = int(raw[1]), int(raw[2])
    return raw[4:].copy(), width, n


def unpack(words, width, n):
    """packed ints -> uint64 array (vectorised; width <= 64)."""
    idx = np.arange(n, dtype=np.uint64) * np.uint64(width)
    wi = (idx >> np.uint64(6)).astype(np.int64)
    sh = idx & np.uint64(63)
    w = np.concatenate([words, np.zeros(2, dtype=np.uint64)])
    lo = w[wi] >> sh
    hi = np.where(sh > 0, w[wi + 1] << ((np.uint64(64) - sh) & np.uint64(63)), np.uint64(0))
    x = lo | hi
    return x if width == 64 else x & np.uint64((1 << width) - 1)


def _read_cereal(f, dtype):
    n = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
    return np.frombuffer(f.read(n * np.dtype(dtype).itemsize), dtype=dtype).copy()


def read_ctable(path):
    """cereal: vec<string> ref names, vec<u32> ref exts, vec<u64> occurrences"""
    with open(path, "rb") as f:
        n = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
        names = []
        for _ in range(n):
            ln = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
            names.append(f.read(ln).decode())
        exts = _read_cereal(f, np.uint32)
        occs = _read_cereal(f, np.uint64)
        assert f.read() == b""
    return names, exts, occs


class DenseIndexFiles:
    """everything DenseIndex::deserialize_from_cpp (src/pf1/dense_index.rs:33-97) reads"""

    def __init__(self, d):
        self.dir = d
        self.info = json.load(open(os.path.join(d, "info.json")))
        self.k = int(self.info["k"])
        seq_words, w, n = read_compact(os.path.join(d, "seq.bin"))
        assert w == 2
        self.useq_words, self.total_len = seq_words, n
        bvw, w, n = read_compact(os.path.join(d, "rank.bin"))
        assert w == 1
        bits = np.unpackbits(bvw.view(np.uint8), bitorder="little")[:n]
        ends = np.nonzero(bits)[0].astype(np.uint64)            # last base of every unitig
        self.starts = np.concatenate([[0], ends + 1]).astype(np.uint64)  # dense_index.rs:55-62
        self.n_unitigs = len(ends)
        pw, w, n = read_compact(os.path.join(d, "pos.bin"))
        self.pos_words, self.pos_width, self.n_pos = pw, w, n
        self.ref_names, self.ref_exts, self.ctable = read_ctable(os.path.join(d, "ctable.bin"))
        cw, w, n = read_compact(os.path.join(d, "ctg_offsets.bin"))
        self.contig_offsets = unpack(cw, w, n)
        with open(os.path.join(d, "refAccumLengths.bin"), "rb") as f:
            acc = _read_cereal(f, np.uint64)
        self.ref_prefix_sum = np.concatenate([[0], acc]).astype(np.uint64)  # pf1/mod.rs:226-234
        rp = os.path.join(d, "refseq.bin")
        if os.path.exists(rp):
            rw, w, n = read_compact(rp)
            assert w == 2
            self.refseq_words, self.refseq_len = rw, n
        else:
            self.refseq_words, self.refseq_len = None, 0
