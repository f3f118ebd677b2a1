#!/usr/bin/env python3
"""One-off stress of the reads entry point on the GPU box: random unitig sets (k = 31, w = 15 -> the wave-scoped reads kernel;
other (k, w) -> the workgroup-staged one) with a planted repeat family, random reads cut across unitig junctions with
substitutions, Ns, lower case, empty / tiny / very long reads; every window against the oracle's reads walk.
MZ_READS_LINES and MZ_DIRECT_BYTES_PER_KEY choose the structures the kernel resolves through."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H
import mazu_amd as M
from oracle import oracle as O

rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "4242")))
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
plant = "--no-plant" not in sys.argv  # without the repeat family the sets have no duplicate k-mers: every window is compared bit for bit
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
bad = 0
for c in range(n_cfg):
    k, w = (31, 15) if rng.random() < 0.7 else (int(rng.choice([31, 27, 21])), int(rng.choice([9, 11, 13])))
    if os.environ.get("STRESS_WIDE"):  # (31, 17 / 19 / 21): the wave-scoped kernel through hash-addressed lines; other w > 16: the workgroup-staged one
        k, w = 31, int(rng.choice([17, 19, 21, 21, 19, 18, 25]))
    skew = int(rng.choice([4, 16, 32]))
    seed = int(rng.integers(0, 1 << 30))
    # small key universes get direct records + lines; (31, 15) sets are forced onto them half of the time
    os.environ.pop("MZ_DIRECT_BYTES_PER_KEY", None)
    os.environ.pop("MZ_READS_LINES", None)
    os.environ.pop("MZ_FRONT_TABLE", None)
    os.environ.pop("MZ_READS_RESOLVE", None)
    mode = int(rng.integers(0, 5))
    if os.environ.get("STRESS_MODE"):  # one form only (e.g. 4: every (31, 15) set through the lines, every window scanning its line)
        mode = int(os.environ["STRESS_MODE"])
    if w == 15 and mode >= 1:
        os.environ["MZ_DIRECT_BYTES_PER_KEY"] = "100000"
        os.environ["MZ_FRONT_TABLE"] = "1"
        os.environ["MZ_READS_LINES"] = "1" if mode >= 2 else "0"
        if mode >= 2:  # through the lines: per run of windows (2, 3) or every window for itself (4)
            os.environ["MZ_READS_RESOLVE"] = "item" if mode == 4 else "run"
    codes, starts = H.random_unitigs(seed, int(rng.integers(100, 1200)), float(rng.choice([60, 200, 800])), k, max_len=4000)
    total = int(starts[-1])
    motif = rng.integers(0, 4, int(rng.integers(k + 5, 120)), dtype=np.uint8)
    for _ in range(int(rng.integers(0, 300)) if plant else 0):
        p = int(rng.integers(0, max(1, total - len(motif))))
        m = motif.copy()
        mut = rng.random(len(m)) < 0.08
        m[mut] = (m[mut] + rng.integers(1, 4, int(mut.sum()))) & 3
        codes[p:p + len(m)] = m
    words = H.pack_codes(codes)
    orc = O.OracleSSHash(words, total, starts, k, w, skew, seed)
    us = M.UnitigSet.from_packed(words, total, starts, k)
    ix = M.SSHash.from_unitig_set_on_device(us, w, skew, seed) if rng.random() < 0.5 else M.SSHash.from_unitig_set(us, w, skew, seed)
    reads = []
    n_reads = int(rng.integers(50, 3000))
    shape = int(rng.integers(0, 4))  # 0: 150 bp, 1: tiny and empty reads mixed in, 2: a few very long reads, 3: ragged
    for r in range(n_reads):
        L = 150 if shape == 0 else int(rng.choice([0, 1, 5, 30, 31, 32, 150])) if shape == 1 else int(rng.choice([150, 5000])) if shape == 2 else int(rng.integers(20, 400))
        a = int(rng.integers(0, max(1, total - L)))
        rd = letters[codes[a:a + L]].copy()
        if len(rd) and rng.random() < 0.5:
            sub = rng.random(len(rd)) < 0.01
            rd[sub] = letters[rng.integers(0, 4, int(sub.sum()))]
        if len(rd) and rng.random() < 0.1:
            rd[int(rng.integers(0, len(rd)))] = ord("N")
        if len(rd) and rng.random() < 0.2:
            rd = np.frombuffer(bytes(rd).lower(), dtype=np.uint8).copy()
        reads.append(rd)
    ro = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in reads], out=ro[1:].view(np.int64))
    bases = np.concatenate(reads) if int(ro[-1]) else np.zeros(0, dtype=np.uint8)
    want, _, oc = orc.k2u_reads(bases, ro, streaming=True)
    got, cnt = ix.k2u_reads(bases, ro, streaming=True)
    if orc.n_duplicate_kmers() == 0:
        ok = got.tobytes() == want.tobytes() and list(cnt) == oc.tolist()
    else:
        ok = np.array_equal(got["match"], np.where(want["match"] == 0, 0, got["match"])) and np.array_equal(got["match"] != 0, want["match"] != 0)
    bad += not ok
    print(f"cfg {c:3d} k={k} w={w} skew={skew} mode={mode} shape={shape} reads={n_reads} bases={int(ro[-1])} lines={ix.has_minimizer_lines()} kind={ix.minimizer_mphf_kind()} "
          f"dups={orc.n_duplicate_kmers()} hits={cnt[0]}: {'ok' if ok else 'FAIL'}", flush=True)
print("FAILED configs:", bad)
sys.exit(1 if bad else 0)
