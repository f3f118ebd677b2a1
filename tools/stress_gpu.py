#!/usr/bin/env python3
"""One-off stress on the GPU box: random unitig sets with planted repeats (heavy buckets -> skew table, negative filter,
overflowed offset blocks) over many (k, w, skew_param, seed); host-built and GPU-built index against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H
import mazu_amd as M
from oracle import oracle as O

rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "12345")))
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for c in range(n_cfg):
    k = int(rng.choice([31, 31, 31, 29, 27, 25, 21, 17, 30, 24]))
    w = int(rng.integers(max(1, k - 22), min(k, 16) + 1)) if rng.random() < 0.8 else int(rng.integers(1, k + 1))
    if os.environ.get("STRESS_WIDE"):  # minimizers beyond 32 bits only: BBHash levels + hash-addressed lines, both builders
        w = int(rng.integers(17, k + 1))
    skew = int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64])) if rng.random() < 0.85 else M.MZ_NO_SKEW_INDEX
    seed = int(rng.integers(0, 1 << 30))
    codes, starts = H.random_unitigs(seed, int(rng.integers(50, 1500)), float(rng.choice([40, 150, 600])), k, max_len=3000)
    total = int(starts[-1])
    # planted repeat family: copies of one motif with 3-20 % substitutions
    motif = rng.integers(0, 4, int(rng.integers(k + 5, 120)), dtype=np.uint8)
    mut_rate = float(rng.choice([0.03, 0.08, 0.2]))
    for _ in range(int(rng.integers(0, 400))):
        p = int(rng.integers(0, max(1, total - len(motif))))
        m = motif.copy()
        mut = rng.random(len(m)) < mut_rate
        m[mut] = (m[mut] + rng.integers(1, 4, int(mut.sum()))) & 3
        codes[p:p + len(m)] = m
    words = H.pack_codes(codes)
    orc = O.OracleSSHash(words, total, starts, k, w, skew, seed)
    us = M.UnitigSet.from_packed(words, total, starts, k)
    q = H.query_mix(words, starts, k, 100000, seed=seed & 0xFFFF)
    pos = q[:30000].copy()
    neg1 = pos ^ (np.uint64(1) << rng.integers(0, 2 * k, len(pos)).astype(np.uint64))
    q = np.concatenate([q, neg1])
    truth, _ = orc.k2u_batch(q, n_threads=8, use_dict=True)
    dups = orc.n_duplicate_kmers()
    files = []
    for how in ("host", "gpu"):
        ix = M.SSHash.from_unitig_set(us, w, skew, seed) if how == "host" else M.SSHash.from_unitig_set_on_device(us, w, skew, seed)
        ix.save("/tmp/stress_%s.mz" % how)
        files.append(open("/tmp/stress_%s.mz" % how, "rb").read())
        if how == "gpu" and files[0] != files[1]:
            bad += 1
            print(f"cfg {c:3d} k={k} w={w}: index files of the two builders DIFFER (built_on_device={ix.built_on_device()})", flush=True)
        got, _ = ix.k2u_batch(q)
        rep = ix.validate_self_report()
        ok = rep["n_fail"] == 0 and rep["n_kmers"] == ix.n_kmers()
        if dups == 0:
            ok = ok and got.tobytes() == truth.tobytes()
        else:
            ok = ok and np.array_equal(got["match"] != 0, truth["match"] != 0)
        if not ok:
            bad += 1
        print(f"cfg {c:3d} k={k} w={w} skew={skew if skew < 1 << 40 else 'none'} unitigs={len(starts)-1} bp={total} dups={dups} "
              f"skew_kmers={ix.n_kmers_in_skew_index()} {how}: {'ok' if ok else 'FAIL'}", flush=True)
print("FAILED configs:", bad)
sys.exit(1 if bad else 0)
