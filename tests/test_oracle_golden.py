"""Pins the CPU oracle (oracle/sshash_oracle.c) against every known-answer test the
reference holds for the k2u path.  CPU only."""
import json
import os

import numpy as np
import pytest

import helpers as H
from oracle import oracle as O

KATS = json.load(open(os.path.join(H.GOLDEN, "reference_kats.json")))


def build(seqs, k, w, skew=O.NO_SKEW, seed=0):
    words, total, starts = H.pack_seqs(seqs)
    return O.OracleSSHash(words, total, starts, k, w, skew, seed)


def check_tiny(ix, kat, use_dict):
    k = kat["k"]
    for hit in kat["hits"]:
        fw = H.encode_kmer(hit["kmer"])
        r = ix.k2u(fw, use_dict)
        assert (r["unitig_id"], r["unitig_len"], r["pos"], r["match"]) == (hit["unitig_id"], hit["unitig_len"], hit["pos"], 1)
        r = ix.k2u(H.rc_word(fw, k), use_dict)  # km.swap()
        assert (r["unitig_id"], r["unitig_len"], r["pos"], r["match"]) == (hit["unitig_id"], hit["unitig_len"], hit["pos"], 2)
    for miss in kat.get("misses", []):
        fw = H.encode_kmer(miss)
        assert ix.k2u(fw, use_dict)["match"] == 0
        assert ix.k2u(H.rc_word(fw, k), use_dict)["match"] == 0


@pytest.mark.parametrize("w", range(1, 8))
def test_tiny_vary_seeds_and_windows(w):  # sshash.rs:765-774
    kat = KATS["tiny_k7"]
    for seed in (0, 1, 7, 99):
        ix = build(kat["unitigs"], kat["k"], w, seed=seed)
        check_tiny(ix, kat, False)
        check_tiny(ix, kat, True)


def test_tiny_validate_self():  # sshash.rs:776-787
    kat = KATS["tiny_k7"]
    ix = build(kat["unitigs"], 7, 3)
    assert ix.validate_self() == 0 and ix.validate_self(use_dict=True) == 0
    ix = build(kat["unitigs"], 7, 5, skew=0)
    assert ix.validate_self() == 0


def test_tiny_skew_index():  # sshash.rs:841-884
    kat = KATS["tiny_k7"]
    sk = build(kat["unitigs"], 7, 3, skew=0)
    no = build(kat["unitigs"], 7, 3)
    assert sk.n_kmers_in_skew_index() == sk.n_kmers() == 8
    for s in ("CACACAC", "ACACCAC", "CCTCAAT"):
        fw = H.encode_kmer(s)
        for q in (fw, H.rc_word(fw, 7)):
            assert sk.k2u(q).tobytes() == no.k2u(q).tobytes()
    assert sk.k2u(H.encode_kmer("AAAAAAA"))["match"] == 0


def test_unitigs_share_mmer():  # sshash.rs:792-824
    kat = KATS["unitigs_share_mmer"]
    ix = build(kat["unitigs"], kat["k"], kat["w"])
    check_tiny(ix, kat, False)
    assert ix.validate_self() == 0


def test_unitig_set_tiny():  # unitig_set.rs:355-381
    kat = KATS["tiny_k7"]
    ix = build(kat["unitigs"], 7, 3)
    assert ix.k() == 7 and ix.n_unitigs() == 2 and ix.sum_unitigs_len() == 20
    assert ix.unitig_len(0) == 10 and ix.unitig_len(1) == 10
    assert all(ix.pos_to_id(i) == 0 for i in range(10))
    assert all(ix.pos_to_id(i) == 1 for i in range(10, 20))


def test_elias_fano_kats():  # elias_fano.rs:148-164
    kat = KATS["elias_fano_vigna"]
    rc, out = O.ef_roundtrip(kat["xs"], kat["u"])
    assert rc == 0 and out.tolist() == kat["xs"]
    kat = KATS["elias_fano_not_monotone"]
    rc, _ = O.ef_roundtrip(kat["xs"], kat["u"])
    assert rc == -1


def test_msb():  # util.rs:42-47
    for n, want in KATS["msb"]["cases"]:
        assert O.msb(n) == want


def test_wrong_w_rejected():  # assert!(w <= k), sshash.rs:92
    with pytest.raises(ValueError):
        build(KATS["tiny_k7"]["unitigs"], 7, 8)


def test_yeast_chr7_validate_self():  # kphf/mod.rs:147-161 and caching.rs:238-253 sanity
    kat = KATS["yeast_chr7"]
    k, seqs = H.load_cf_seqs(os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7"))
    assert k == kat["k"] and len(seqs) == kat["n_unitigs"]
    ix = build(seqs, k, 15)
    assert ix.sum_unitigs_len() == kat["total_len"] and ix.n_kmers() == kat["n_kmers"]
    assert ix.n_duplicate_kmers() == 0
    assert ix.validate_self(use_dict=True) == 0
    assert ix.validate_self() == 0
    sk = build(seqs, k, 15, skew=32)
    assert sk.n_minimizers() < sk.n_kmers()
    assert sk.n_kmers_in_skew_index() != 0
    assert sk.validate_self() == 0
    # restatement == brute force on a 50/50 mix, and skew == no-skew
    words, total, starts = H.pack_seqs(seqs)
    q = H.query_mix(words, starts, k, 200000, seed=5)
    a, nh = sk.k2u_batch(q, n_threads=4)
    b, nh2 = sk.k2u_batch(q, n_threads=4, use_dict=True)
    c, _ = ix.k2u_batch(q, n_threads=4)
    assert nh == nh2 and a.tobytes() == b.tobytes() == c.tobytes()
    assert 0.45 < nh / len(q) < 0.55


def test_reads_streaming_equals_cold():  # caching.rs:65-103; bin/kphf/main.rs:299-324
    k, seqs = H.load_cf_seqs(os.path.join(H.GOLDEN, "cf", "tiny", "tiny"))
    ix = build(seqs, k, 3)
    recs = H.read_fasta(os.path.join(H.GOLDEN, "cf", "tiny", "tiny.fa"))
    bases = np.frombuffer("".join(s for _, s in recs).encode(), dtype=np.uint8)
    ro = np.cumsum([0] + [len(s) for _, s in recs]).astype(np.uint64)
    cold, ko, cc = ix.k2u_reads(bases, ro, streaming=False)
    warm, _, cw = ix.k2u_reads(bases, ro, streaming=True)
    assert cold.tobytes() == warm.tobytes() and cc.tolist() == cw.tolist()
    # tiny.fa: NNNCACACACCACNCGTATTGAGG -> 4 k-mers of unitig 0 fw, then 4 of unitig 1 rc
    r0 = cold[: int(ko[1])]
    valid = r0[r0["match"] != 3]
    assert len(valid) == 8 and (valid["match"][:4] == 1).all() and (valid["match"][4:] == 2).all()
    assert valid["unitig_id"].tolist() == [0, 0, 0, 0, 1, 1, 1, 1]
    assert valid["pos"].tolist() == [0, 1, 2, 3, 3, 2, 1, 0]
