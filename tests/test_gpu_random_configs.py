"""Seeded sweep over (k, w, skew_param, unitig shapes) on the GPU: random unitig sets with a planted repeat family (heavy
buckets -> skew table, negative filter, overflowed offset blocks, spilled minimizer lines), index built by the host builder
and by the GPU builder, every answer compared with the oracle's brute-force dictionary (positives, NEG-U and NEG-1
negatives) and the index checked by the device validate_self (reference src/kphf/mod.rs:69-139)."""
import numpy as np
import pytest

import helpers as H
import mazu_amd as M
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _one_config(rng):
    k = int(rng.choice([31, 31, 31, 29, 27, 25, 21, 17, 30, 24]))
    w = int(rng.integers(max(1, k - 22), min(k, 16) + 1)) if rng.random() < 0.8 else int(rng.integers(1, k + 1))
    skew = int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64])) if rng.random() < 0.85 else M.MZ_NO_SKEW_INDEX
    seed = int(rng.integers(0, 1 << 30))
    codes, starts = H.random_unitigs(seed, int(rng.integers(50, 800)), float(rng.choice([40, 150, 600])), k, max_len=3000)
    total = int(starts[-1])
    motif = rng.integers(0, 4, int(rng.integers(k + 5, 120)), dtype=np.uint8)  # repeat family: copies with 3-20 % substitutions
    mut_rate = float(rng.choice([0.03, 0.08, 0.2]))
    for _ in range(int(rng.integers(0, 300))):
        p = int(rng.integers(0, max(1, total - len(motif))))
        m = motif.copy()
        mut = rng.random(len(m)) < mut_rate
        m[mut] = (m[mut] + rng.integers(1, 4, int(mut.sum()))) & 3
        codes[p:p + len(m)] = m
    return k, w, skew, seed, codes, starts, total


@pytest.mark.parametrize("block", range(4))
def test_random_configurations_match_the_dictionary(block):
    rng = np.random.default_rng(2024 + block)
    for c in range(12):
        k, w, skew, seed, codes, starts, total = _one_config(rng)
        words = H.pack_codes(codes)
        orc = O.OracleSSHash(words, total, starts, k, w, skew, seed)
        us = M.UnitigSet.from_packed(words, total, starts, k)
        q = H.query_mix(words, starts, k, 40000, seed=seed & 0xFFFF)
        pos = q[:10000].copy()
        neg1 = pos ^ (np.uint64(1) << rng.integers(0, 2 * k, len(pos)).astype(np.uint64))  # one substituted base
        q = np.concatenate([q, neg1])
        truth, _ = orc.k2u_batch(q, n_threads=8, use_dict=True)
        dups = orc.n_duplicate_kmers()
        for how in ("host", "gpu"):
            ix = M.SSHash.from_unitig_set(us, w, skew, seed) if how == "host" else M.SSHash.from_unitig_set_on_device(us, w, skew, seed)
            got, _ = ix.k2u_batch(q)
            rep = ix.validate_self_report()
            what = f"block {block} cfg {c}: k={k} w={w} skew={skew} seed={seed} bp={total} dups={dups} {how}"
            if how == "gpu" and w > 16:
                assert ix.built_on_device(), what  # minimizers beyond 32 bits: BBHash levels and hash-addressed lines made on the device
            assert rep["n_fail"] == 0 and rep["n_kmers"] == ix.n_kmers(), what
            if dups == 0:
                assert got.tobytes() == truth.tobytes(), what
            else:  # duplicate k-mers (outside the parity contract): which occurrence answers is not fixed, hit / miss is
                assert np.array_equal(got["match"] != 0, truth["match"] != 0), what


def test_folded_lines_retry_when_a_line_overflows(tmp_path):
    """A minimizer-line table folded so hard that some line would hold more than 16 member keys is rebuilt with less folding,
    identically by the host and the GPU builder (the flat files are compared byte for byte), and answers like the dictionary."""
    # mz_build_opts.lines_max_load = 1000: start from the fewest lines the slot's key field allows
    rng = np.random.default_rng(77)
    for c, (k, w) in enumerate([(31, 11), (25, 9), (31, 12)]):
        seed = int(rng.integers(0, 1 << 30))
        codes, starts = H.random_unitigs(seed, 2500, 300.0, k, max_len=3000)
        total = int(starts[-1])
        words = H.pack_codes(codes)
        orc = O.OracleSSHash(words, total, starts, k, w, 32, seed)
        us = M.UnitigSet.from_packed(words, total, starts, k)
        q = H.query_mix(words, starts, k, 40000, seed=seed & 0xFFFF)
        truth, _ = orc.k2u_batch(q, n_threads=8, use_dict=True)
        files = []
        for how in ("host", "gpu"):
            ix = M.SSHash.from_unitig_set_opts(us, w, 32, seed, device=-1 if how == "host" else 0, lines_max_load=1000)
            assert ix.has_minimizer_lines(), (c, how)
            got, _ = ix.k2u_batch(q)
            rep = ix.validate_self_report()
            assert rep["n_fail"] == 0 and rep["n_kmers"] == ix.n_kmers(), (c, how)
            if orc.n_duplicate_kmers() == 0:
                assert got.tobytes() == truth.tobytes(), (c, how)
            path = tmp_path / f"{c}_{how}.mz"
            ix.save(path)
            files.append(open(path, "rb").read())
        assert files[0] == files[1], c
        if c == 1:  # 4^9 words over 1024 lines at full folding: the builders had to back off
            from test_gpu_configs import FlatIndexFile
            f = FlatIndexFile(files[0])
            assert f.has_front and f.front_g + f.front_t < 8 and f.n("front_lines") == 16 * 4 ** w >> (f.front_g + f.front_t)


@pytest.mark.parametrize("reads_lines", ["lines", "records"])
def test_bucket_larger_than_a_marker_can_say_without_a_skew_index(reads_lines):
    """90 000 unitigs that all carry one 21-base motif between unique flanks: without a skew index their shared minimizer owns a
    bucket whose size does not fit the 16-bit count of a minimizer-line spill marker, so lookups through the lines (point queries,
    and reads resolved per run of windows) must fall back to the exact in-order walk of sshash.rs:494-553."""
    rng = np.random.default_rng(5)
    k, w, n = 31, 11, 90000
    motif = rng.integers(0, 4, 21, dtype=np.uint8)
    idx = np.arange(n, dtype=np.int64)
    flank = np.stack([(idx >> (2 * j)) & 3 for j in range(10)], axis=1).astype(np.uint8)  # unitig number in base 4: unique 31-mers
    codes = np.concatenate([flank[:, :5], np.broadcast_to(motif, (n, 21)), flank[:, 5:]], axis=1).reshape(-1)
    # plus ordinary random unitigs, so that the set has light buckets as well
    extra, extra_starts = H.random_unitigs(9, 400, 200.0, k, max_len=2000)
    starts = np.concatenate([np.arange(n, dtype=np.uint64) * 31, np.uint64(n * 31) + extra_starts])
    codes = np.concatenate([codes, extra])
    total = int(starts[-1])
    words = H.pack_codes(codes)
    orc = O.OracleSSHash(words, total, starts, k, w, M.MZ_NO_SKEW_INDEX, 0)
    us = M.UnitigSet.from_packed(words, total, starts, k)
    ix = M.SSHash.from_unitig_set_no_skew_index(us, w)
    ix.set_query_opts(reads_lines=reads_lines)
    assert ix.has_minimizer_lines() and ix.minimizer_mphf_kind() == 1
    q = H.query_mix(words, starts, k, 60000, seed=3)
    truth, _ = orc.k2u_batch(q, n_threads=8, use_dict=True)
    got, _ = ix.k2u_batch(q)
    rep = ix.validate_self_report()
    assert rep["n_fail"] == 0 and rep["n_kmers"] == ix.n_kmers()
    if orc.n_duplicate_kmers() == 0:
        assert got.tobytes() == truth.tobytes()
    else:
        assert np.array_equal(got["match"] != 0, truth["match"] != 0)
    # reads: the first 3 000 motif unitigs as 31-base reads (one window each) and 300 reads cut from the random unitigs
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [letters[codes[31 * i:31 * i + 31]] for i in range(3000)]
    for j in range(300):
        a = int(starts[n + j]); b = int(starts[n + j + 1])
        reads.append(letters[codes[a:min(b, a + 150)]])
    ro = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in reads], out=ro[1:].view(np.int64))
    bases = np.concatenate(reads)
    want, _, oc = orc.k2u_reads(bases, ro, streaming=True)
    got_r, cnt = ix.k2u_reads(bases, ro, streaming=True)
    if orc.n_duplicate_kmers() == 0:
        assert got_r.tobytes() == want.tobytes() and list(cnt) == oc.tolist()
    else:
        assert np.array_equal(got_r["match"] != 0, want["match"] != 0)
    assert cnt[0] >= 3000


@pytest.mark.parametrize("k,w,load", [(31, 17, 0), (31, 19, 1000), (31, 21, 2), (27, 18, 40), (63 // 2, 25, 0)])
def test_hash_addressed_lines_answer_like_the_index_without_them(k, w, load, tmp_path):
    """2w > 32: the minimizer lines are addressed by a hash of the word and carry an 8-bit fingerprint; lines with a fingerprint clash or more
    than 16 words are given up and their queries take the MPHF path.  Whatever the load (lines_max_load = 1000 folds everything into
    a few lines: all given up; 2: sparse), the answers are those of the oracle's dictionary and of the same index without lines,
    also after a save / load round trip."""
    rng = np.random.default_rng(k * 100 + w)
    seed = int(rng.integers(0, 1 << 30))
    codes, starts = H.random_unitigs(seed, 3000, 250.0, k, max_len=3000)
    total = int(starts[-1])
    motif = rng.integers(0, 4, 90, dtype=np.uint8)  # a repeat family: heavy buckets, spilled lines
    for _ in range(400):
        p = int(rng.integers(0, total - len(motif)))
        m = motif.copy()
        mut = rng.random(len(m)) < 0.08
        m[mut] = (m[mut] + rng.integers(1, 4, int(mut.sum()))) & 3
        codes[p:p + len(m)] = m
    words = H.pack_codes(codes)
    orc = O.OracleSSHash(words, total, starts, k, w, 8, seed)
    us = M.UnitigSet.from_packed(words, total, starts, k)
    q = H.query_mix(words, starts, k, 60000, seed=7)
    neg1 = q[:20000] ^ (np.uint64(1) << rng.integers(0, 2 * k, 20000).astype(np.uint64))
    q = np.concatenate([q, neg1])
    truth, _ = orc.k2u_batch(q, n_threads=8, use_dict=True)
    plain = M.SSHash.from_unitig_set_opts(us, w, 8, seed, minimizer_lines=-1)
    assert not plain.has_minimizer_lines()
    want, _ = plain.k2u_batch(q)
    ix = M.SSHash.from_unitig_set_opts(us, w, 8, seed, lines_max_load=load)
    st = ix.line_stats()
    assert ix.has_minimizer_lines() and st["g"] == 0 and st["n_lines"] == 1 << st["t"]
    if load == 1000:
        assert st["lines_unusable"] == st["n_lines"] and st["entries_inline"] == 0
    elif load == 40:  # ~25 words per line: hardly a line survives, the few that do must still answer
        assert 0.9 * st["n_lines"] < st["lines_unusable"] < st["n_lines"]
    else:
        assert st["lines_unusable"] < 0.1 * st["n_lines"] and st["frac_inline"] > 0.7
    ix.save(tmp_path / "h.mz")
    for cand in (ix, M.SSHash.load(tmp_path / "h.mz")):
        got, _ = cand.k2u_batch(q)
        assert got.tobytes() == want.tobytes()
        rep = cand.validate_self_report()
        assert rep["n_fail"] == 0 and rep["n_kmers"] == cand.n_kmers()
    if orc.n_duplicate_kmers() == 0:
        assert want.tobytes() == truth.tobytes()
    else:
        assert np.array_equal(want["match"] != 0, truth["match"] != 0)
