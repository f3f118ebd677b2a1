"""GPU parity tests (run on the MI355X box with -m gpu): every call goes through the C ABI of
include/mazu_amd.h into the HIP kernels and is compared bit-for-bit with the CPU oracle, the
brute-force dictionary, the reference's known-answer vectors, or a size-independent property."""
import json
import os

import numpy as np
import pytest

import helpers as H
import mazu_amd as M
from oracle import oracle as O

pytestmark = pytest.mark.gpu

KATS = json.load(open(os.path.join(H.GOLDEN, "reference_kats.json")))
CF_TINY = os.path.join(H.GOLDEN, "cf", "tiny", "tiny")
CF_YEAST = os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7")


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    assert M.lib.mz_device_count() >= 1, "GPU tests need a visible HIP device (no CPU fallback exists)"


def both(seqs, k, w, skew=M.MZ_NO_SKEW_INDEX, seed=0):
    words, total, starts = H.pack_seqs(seqs)
    ix = M.SSHash.from_unitig_set(M.UnitigSet.from_seqs(seqs, k), w, skew, seed)
    orc = O.OracleSSHash(words, total, starts, k, w, skew, seed)
    return ix, orc, (words, total, starts)


def check_kat(ix, kat):
    for hit in kat["hits"]:
        km = M.CanonicalKmer(hit["kmer"])
        ans = M.K2UPos(hit["unitig_id"], hit["unitig_len"], hit["pos"], M.MatchType.IdentityMatch)
        assert ix.k2u(km) == ans
        km.swap()
        ans.reverse_match_type()
        assert ix.k2u(km) == ans
    for miss in kat.get("misses", []):
        km = M.CanonicalKmer(miss)
        assert ix.k2u(km) is None
        km.swap()
        assert ix.k2u(km) is None


@pytest.mark.parametrize("w", range(1, 8))
def test_tiny_vary_seeds_and_windows(w):  # sshash.rs:645-774
    kat = KATS["tiny_k7"]
    us = M.UnitigSet.from_cf_prefix(CF_TINY)
    for seed in (0, 1, 42):
        check_kat(M.SSHash.from_unitig_set_no_skew_index(us, w, seed), kat)


def test_tiny_validate_self():  # sshash.rs:776-787
    us = M.UnitigSet.from_cf_prefix(CF_TINY)
    M.SSHash.from_unitig_set_no_skew_index(us, 3).validate_self()
    M.SSHash.from_unitig_set(us, 5, 0).validate_self()


def test_tiny_skew_index():  # sshash.rs:841-884
    us = M.UnitigSet.from_cf_prefix(CF_TINY)
    sk, no = M.SSHash.from_unitig_set(us, 3, 0), M.SSHash.from_unitig_set_no_skew_index(us, 3)
    assert sk.n_kmers_in_skew_index() == sk.n_kmers()
    assert sk.num_bits() > no.num_bits()
    for s in ("CACACAC", "ACACCAC", "CCTCAAT", "CCTCAAT"):
        km = M.CanonicalKmer(s)
        assert sk.k2u(km) == no.k2u(km) and sk.k2u(km) is not None
        km.swap()
        assert sk.k2u(km) == no.k2u(km)
    assert sk.k2u(M.CanonicalKmer("AAAAAAA")) is None


def test_unitigs_share_mmer():  # sshash.rs:792-824
    kat = KATS["unitigs_share_mmer"]
    ix = M.SSHash.from_unitig_set_no_skew_index(M.UnitigSet.from_seqs(kat["unitigs"], 31), 15)
    check_kat(ix, kat)
    ix.validate_self()


def test_wrong_k_is_an_error():  # sshash.rs:826-838 (#[should_panic]) -> MZ_EINVAL
    ix = M.SSHash.from_unitig_set_no_skew_index(M.UnitigSet.from_cf_prefix(CF_TINY), 3)
    with pytest.raises(M.MazuError) as e:
        ix.k2u(M.CanonicalKmer.from_u64(0, ix.k() + 1))
    assert e.value.code == M._capi.MZ_EINVAL
    with pytest.raises(M.MazuError):
        ix.as_streaming().k2u_streaming(M.CanonicalKmer("ACG"))


@pytest.fixture(scope="module")
def yeast():
    k, seqs = H.load_cf_seqs(CF_YEAST)
    return k, seqs


@pytest.mark.parametrize("w,skew", [(15, M.MZ_NO_SKEW_INDEX), (15, 32), (15, 16), (11, 8), (20, 2), (31, M.MZ_NO_SKEW_INDEX), (1, 64), (4, M.MZ_NO_SKEW_INDEX),
                                    (17, 32), (19, 16), (21, M.MZ_NO_SKEW_INDEX)])
def test_yeast_validate_self_and_oracle_parity(yeast, w, skew):  # kphf/mod.rs:147-161
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, w, skew)
    if w > 16:  # minimizer words beyond 32 bits: lines addressed by a hash of the word, the BBHash levels behind the few lines given up
        st = ix.line_stats()
        assert ix.has_minimizer_lines() and ix.minimizer_mphf_kind() == 0 and st["g"] == 0 and st["n_lines"] == 1 << st["t"]
        assert st["lines_unusable"] < 0.01 * st["n_lines"] and 0 < st["buckets_walk"] < 0.01 * ix.n_minimizers() and st["frac_inline"] > 0.9
    if (w, skew) == (15, 16):
        assert ix.n_kmers_in_skew_index() != 0 and ix.n_minimizers() < ix.n_kmers()  # caching.rs:247-248 (see test_boundary)
    rep = ix.validate_self_report()
    assert rep == {"n_kmers": 1071346, "n_fail": 0, "n_duplicate": 0, "first_fail_pos": 2**64 - 1}
    q = H.query_mix(words, starts, k, 300000, seed=w)
    rng = np.random.default_rng(w)
    neg1 = q[:50000] ^ (np.uint64(1) << rng.integers(0, 62, 50000).astype(np.uint64))  # NEG-1
    q = np.concatenate([q, neg1])
    got, counts = ix.k2u_batch(q)
    want, n_hit = orc.k2u_batch(q, n_threads=8)
    truth, _ = orc.k2u_batch(q, n_threads=8, use_dict=True)
    assert want.tobytes() == truth.tobytes()
    assert got.tobytes() == want.tobytes()
    assert counts == (n_hit, len(q) - n_hit, 0)


@pytest.mark.parametrize("k,w,skew,seed", [(31, 15, 4, 1), (21, 11, 3, 2), (15, 7, 2, 3), (9, 4, 1, 4), (5, 3, 0, 5), (3, 2, M.MZ_NO_SKEW_INDEX, 6),
                                           (31, 31, M.MZ_NO_SKEW_INDEX, 7), (27, 1, 16, 8)])
def test_random_unitig_sets_match_oracle(k, w, skew, seed):
    codes, starts = H.random_unitigs(seed, 400, 150 if k > 9 else 3, k, max_len=800)
    # (small k: random sequence repeats k-mers; which occurrence answers is out of the parity contract, so the branch for sets
    # with duplicates below compares hit / miss with the dictionary and lets validate_self count the duplicates)
    words = H.pack_codes(codes)
    total = int(starts[-1])
    orc = O.OracleSSHash(words, total, starts, k, w, skew, seed)
    ix = M.SSHash.from_unitig_set(M.UnitigSet.from_packed(words, total, starts, k), w, skew, seed)
    q = H.query_mix(words, starts, k, 40000, seed=seed)
    got, _ = ix.k2u_batch(q)
    truth, _ = orc.k2u_batch(q, n_threads=4, use_dict=True)
    if orc.n_duplicate_kmers() == 0:
        want, _ = orc.k2u_batch(q, n_threads=4)
        assert got.tobytes() == want.tobytes() == truth.tobytes()
        rep = ix.validate_self_report()
        assert rep["n_fail"] == 0 and rep["n_duplicate"] == 0 and rep["n_kmers"] == ix.n_kmers()
    else:  # duplicates present: hit/miss and the matched k-mer must still agree with the dictionary
        assert np.array_equal(got["match"] != 0, truth["match"] != 0)
        rep = ix.validate_self_report()
        assert rep["n_fail"] == 0 and rep["n_duplicate"] > 0


def test_edge_batches(yeast):
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, 15, 32)
    got, counts = ix.k2u_batch(np.zeros(0, dtype=np.uint64))  # empty
    assert len(got) == 0 and counts == (0, 0, 0)
    # first / last k-mer of every unitig and the k-1 junction windows between unitigs (must miss)
    s = starts.astype(np.int64)
    firsts, lasts = s[:-1], s[1:] - k
    junction = np.concatenate([np.arange(e - k + 1, e) for e in s[1:-1]])
    pos = np.concatenate([firsts, lasts, junction])
    q = H.kmer_words_at(words, pos, k)
    got, _ = ix.k2u_batch(q)
    truth, _ = orc.k2u_batch(q, use_dict=True)
    assert got.tobytes() == truth.tobytes()
    n_in = len(firsts) + len(lasts)
    assert (got["match"][:n_in] == 1).all()
    assert (got["pos"][: len(firsts)] == 0).all()
    # single query, all-A / all-T extremes, garbage high bits are masked
    for word in (0, (1 << 62) - 1, int(q[0]) | (3 << 62)):
        r, _ = ix.k2u_batch(np.array([word], dtype=np.uint64))
        assert r.tobytes() == orc.k2u_batch(np.array([word & ((1 << 62) - 1)], dtype=np.uint64), use_dict=True)[0].tobytes()


def test_rc_symmetry_and_idempotence(yeast):
    """size-independent properties: k2u(rc(q)) is k2u(q) with the match type reversed; repeated calls agree."""
    k, seqs = yeast
    ix, _, (words, total, starts) = both(seqs, k, 15, 32)
    q = H.query_mix(words, starts, k, 200000, seed=9)
    a, ca = ix.k2u_batch(q)
    b, cb = ix.k2u_batch(H.rc_words(q, k))
    a2, _ = ix.k2u_batch(q)
    assert a.tobytes() == a2.tobytes() and ca == cb
    for f in ("unitig_id", "pos", "unitig_len"):
        assert np.array_equal(a[f], b[f])
    flip = np.array([0, 2, 1, 3], dtype=np.uint32)
    assert np.array_equal(flip[a["match"]], b["match"])


def reads_fixture(recs):
    bases = np.frombuffer("".join(s for _, s in recs).encode(), dtype=np.uint8)
    ro = np.cumsum([0] + [len(s) for _, s in recs]).astype(np.uint64)
    return bases, ro


@pytest.mark.parametrize("streaming", [False, True])
def test_reads_tiny_fasta_with_Ns(streaming):  # piscem_index.rs:63-99 drives tiny.fa (contains N) through streaming
    k, seqs = H.load_cf_seqs(CF_TINY)
    ix, orc, _ = both(seqs, k, 3)
    bases, ro = reads_fixture(H.read_fasta(CF_TINY + ".fa"))
    got, counts = ix.k2u_reads(bases, ro, streaming=streaming)
    want, _, oc = orc.k2u_reads(bases, ro, streaming=streaming)
    assert got.tobytes() == want.tobytes() and list(counts) == oc.tolist()
    assert counts[0] == 16 and counts[1] == 0


@pytest.mark.parametrize("streaming", [False, True])
def test_reads_yeast_reference_and_noisy_reads(yeast, streaming):  # caching.rs:238-253, bin/kphf/main.rs:299-324
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, 15, 32)
    recs = H.read_fasta(CF_YEAST + ".fa")
    ref = recs[0][1][:300000]
    rng = np.random.default_rng(5)
    reads = [ref]                                    # one long record: every k-mer is a true positive
    for _ in range(400):                             # ragged short reads incl. shorter than k and with N / errors
        n = int(rng.integers(1, 260))
        s = int(rng.integers(0, len(ref) - n))
        r = list(ref[s:s + n])
        for j in range(n):
            u = rng.random()
            if u < 0.01:
                r[j] = "ACGT"[int(rng.integers(0, 4))]
            elif u < 0.013:
                r[j] = "N"
            elif u < 0.02:
                r[j] = r[j].lower()
        reads.append("".join(r))
    reads.append("")
    bases, ro = reads_fixture([("", r) for r in reads])
    got, counts = ix.k2u_reads(bases, ro, streaming=streaming)
    want, ko, oc = orc.k2u_reads(bases, ro, streaming=False)
    assert got.tobytes() == want.tobytes() and list(counts) == oc.tolist()
    assert (got["match"][: int(ko[1])] != 0).all()   # validate_fasta property: every reference k-mer is found


def test_streaming_k2u_matches_oracle_walk(yeast):  # caching.rs:65-103
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, 15, 32)
    ref = H.read_fasta(CF_YEAST + ".fa")[0][1][5000:5400]
    bases, ro = reads_fixture([("", ref)])
    want, _, _ = orc.k2u_reads(bases, ro, streaming=True)
    st = ix.as_streaming()
    assert st.k() == 31
    for i in range(len(ref) - k + 1):
        r = st.k2u_streaming(M.CanonicalKmer(ref[i:i + k]))
        w_ = M.K2UPos.from_record(want[i])
        assert r == w_, (i, r, w_)


def test_device_pointer_and_compact_entry_points(yeast):
    import torch
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, 15, 32)
    ix.to_device(0)
    q = H.query_mix(words, starts, k, 250000, seed=21)
    dq = torch.from_numpy(q.view(np.int64)).cuda()
    dout = torch.zeros(len(q) * 2, dtype=torch.int64, device="cuda")
    dcnt = torch.zeros(3, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ix.k2u_batch_device(dq, dout, dcnt, stream)
    ix.k2u_batch_device(dq, dout, dcnt, stream)  # counts accumulate
    torch.cuda.synchronize()
    got = dout.cpu().numpy().view(M.K2UPOS_DTYPE)
    want, n_hit = orc.k2u_batch(q, n_threads=8)
    assert got.tobytes() == want.tobytes()
    assert dcnt.cpu().tolist() == [2 * n_hit, 2 * (len(q) - n_hit), 0]
    # compacted hits (wave ballot compaction): same set of (query index, K2UPos)
    hidx = torch.zeros(len(q), dtype=torch.int64, device="cuda")
    hits = torch.zeros(len(q) * 2, dtype=torch.int64, device="cuda")
    nh = torch.zeros(1, dtype=torch.int64, device="cuda")
    ix.k2u_batch_compact_device(dq, hidx, hits, nh, stream)
    torch.cuda.synchronize()
    n = int(nh.item())
    assert n == n_hit
    idx = hidx[:n].cpu().numpy()
    rec = hits[: 2 * n].cpu().numpy().view(M.K2UPOS_DTYPE)
    order = np.argsort(idx)
    assert np.array_equal(idx[order], np.nonzero(want["match"])[0])
    assert rec[order].tobytes() == want[want["match"] != 0].tobytes()
    # positives synthesised on the device are what the host extracts
    pos = torch.from_numpy(H.valid_kmer_positions(starts, k)[::997].copy()).cuda()
    flip = (torch.arange(pos.numel(), device="cuda") % 2).to(torch.uint8)
    fw = torch.zeros(pos.numel(), dtype=torch.int64, device="cuda")
    ix.extract_kmers_device(pos, flip, fw, stream)
    torch.cuda.synchronize()
    hw = H.kmer_words_at(words, pos.cpu().numpy(), k)
    hw = np.where(flip.cpu().numpy() != 0, H.rc_words(hw, k), hw)
    assert np.array_equal(fw.cpu().numpy().view(np.uint64), hw)


def test_save_load_roundtrip_on_gpu(yeast, tmp_path):
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, 15, 32)
    ix.save(tmp_path / "y.mz")
    jx = M.SSHash.load(tmp_path / "y.mz")
    q = H.query_mix(words, starts, k, 50000, seed=2)
    assert ix.k2u_batch(q)[0].tobytes() == jx.k2u_batch(q)[0].tobytes()
    jx.validate_self()


def test_synthetic_s_yeast_scale_properties():
    """BASELINE config[1] at its size (12 Mbp, planted minimizer skew, ONE batch of 10 M queries): validate_self on the device,
    50/50 mix hit rate, and a 200k sample of the batch against the oracle."""
    import torch
    from mazu_amd import synth
    us = synth.make_unitig_set("s-yeast", device="cuda")
    words = us["words"].cpu().numpy().view(np.uint64)
    starts = us["starts"].cpu().numpy().view(np.uint64)
    ix = M.SSHash.from_unitig_set(M.UnitigSet.from_packed(words, us["total_len"], starts, 31), 15, 32)
    ix.to_device(0)
    assert ix.n_kmers_in_skew_index() > 0
    rep = ix.validate_self_report()
    assert rep["n_fail"] == 0 and rep["n_kmers"] == ix.n_kmers() and rep["n_duplicate"] <= 4
    q = synth.make_query_mix(ix, us, 10_000_000, seed=3, device="cuda")
    out = torch.zeros(q.numel() * 2, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
    ix.k2u_batch_device(q, out, cnt, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n_hit, n_miss, _ = cnt.cpu().tolist()
    assert n_hit + n_miss == q.numel() and abs(n_hit - q.numel() // 2) <= 40  # NEG-U words that happen to be members
    orc = O.OracleSSHash(words, us["total_len"], starts, 31, 15, 32, with_dict=False)
    qs = q[:200000].cpu().numpy().view(np.uint64)
    want, _ = orc.k2u_batch(qs, n_threads=8)
    assert out[:400000].cpu().numpy().view(M.K2UPOS_DTYPE).tobytes() == want.tobytes()


@pytest.mark.parametrize("n", [1, 63, 64, 1023, 1024, 1025, 4097])
def test_batch_sizes_around_tile_boundaries(yeast, n):
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, 15, 16)
    q = H.query_mix(words, starts, k, n, seed=n)
    got, counts = ix.k2u_batch(q)
    want, n_hit = orc.k2u_batch(q)
    assert got.tobytes() == want.tobytes() and counts == (n_hit, n - n_hit, 0)


def test_degenerate_unitig_sets():
    """unitigs shorter than k (hold no k-mer), exactly k long, one single unitig, w == k"""
    rng = np.random.default_rng(12)
    def rnd(n): return "".join("ACGT"[int(c)] for c in rng.integers(0, 4, n))
    seqs = [rnd(31), rnd(5), rnd(400), rnd(30), rnd(31), rnd(1), rnd(1200), rnd(64)]
    for w, skew in ((15, M.MZ_NO_SKEW_INDEX), (31, M.MZ_NO_SKEW_INDEX), (8, 1), (19, 0)):
        ix, orc, (words, total, starts) = both(seqs, 31, w, skew)
        assert ix.n_kmers() == orc.n_kmers()  # n_kmers counts per the formula of unitig_set.rs:212-214
        q = H.query_mix(words, starts, 31, 5000, seed=w)
        # every k-mer start position in the concatenation, incl. the ones straddling two unitigs
        allpos = np.arange(0, total - 31 + 1)
        q = np.concatenate([q, H.kmer_words_at(words, allpos, 31)])
        got, _ = ix.k2u_batch(q)
        truth, _ = orc.k2u_batch(q, use_dict=True)
        want, _ = orc.k2u_batch(q)
        assert want.tobytes() == truth.tobytes() and got.tobytes() == want.tobytes()
    one = [rnd(5000)]
    ix, orc, (words, total, starts) = both(one, 21, 11, 4)
    rep = ix.validate_self_report()
    assert rep["n_fail"] == 0 and rep["n_kmers"] == 5000 - 21 + 1


def test_reads_edge_shapes(yeast):
    """reads exactly k long, k-1 long, empty, all-N, and a batch whose bases end exactly on a tile boundary"""
    k, seqs = yeast
    ix, orc, _ = both(seqs, k, 15, 16)
    ref = H.read_fasta(CF_YEAST + ".fa")[0][1]
    reads = [ref[100:100 + k], ref[200:200 + k - 1], "", "N" * 40, ref[1000:1000 + 1024 - (k + k - 1 + 40)], ref[5000:5000 + 2048]]
    bases, ro = reads_fixture([("", r) for r in reads])
    got, counts = ix.k2u_reads(bases, ro, streaming=True)
    want, ko, oc = orc.k2u_reads(bases, ro, streaming=True)
    assert got.tobytes() == want.tobytes() and list(counts) == oc.tolist()
    assert counts[2] == 40 - k + 1


@pytest.mark.parametrize("w,skew", [(15, 32), (12, 16)])  # BBHash levels / direct-address minimizer records
def test_reads_crowded_and_very_long(yeast, w, skew):
    """tiles of the reads kernel touched by more reads than it stages (runs of 16-base, 1-base and empty reads: 64 and more read
    starts per 1024 positions), reads longer than several tiles, and reads whose ends fall on tile boundaries, in one batch:
    every window equals the oracle's, and read bookkeeping carried from tile to tile stays right after each kind of tile"""
    k, seqs = yeast
    ix, orc, _ = both(seqs, k, w, skew)
    ref = H.read_fasta(CF_YEAST + ".fa")[0][1]
    rng = np.random.default_rng(11)
    reads, at = [], 1000

    def take(n):
        nonlocal at
        at += int(rng.integers(0, 50))
        r = ref[at:at + n]
        at += n
        return r
    for block in range(6):
        reads += [take(16) for _ in range(200)]                       # exactly 64 reads per 1024 bases
        reads.append(take(3000))                                      # spans three tiles and more
        reads += [take(int(rng.integers(0, 4))) for _ in range(300)]  # empty and 1..3-base reads: hundreds per tile
        reads.append(take(k + int(rng.integers(0, 10))))
        reads += [take(int(rng.choice([0, 5, 31, 33, 40, 64, 200]))) for _ in range(150)]
        pad = (-sum(len(r) for r in reads)) % 1024                    # the next read starts exactly on a tile boundary
        reads.append(take(pad if pad >= k else pad + 1024))
        reads.append(take(1024))                                      # exactly one tile
    bases, ro = reads_fixture([("", r) for r in reads])
    got, counts = ix.k2u_reads(bases, ro, streaming=True)
    want, ko, oc = orc.k2u_reads(bases, ro, streaming=True)
    assert got.tobytes() == want.tobytes() and list(counts) == oc.tolist()
    assert counts[0] > 0.9 * int(ko[-1])                              # slices of the reference: nearly every window is a hit


@pytest.mark.parametrize("prefix,w,skew", [(CF_TINY, 3, M.MZ_NO_SKEW_INDEX), (CF_TINY, 5, 0), (CF_TINY, 1, 2), (CF_TINY, 7, M.MZ_NO_SKEW_INDEX),
                                           (CF_YEAST, 12, 16), (CF_YEAST, 13, 32), (CF_YEAST, 9, 4), (CF_YEAST, 4, M.MZ_NO_SKEW_INDEX),
                                           (CF_YEAST, 15, 16), (CF_YEAST, 20, 8), (CF_YEAST, 17, 32), (CF_YEAST, 19, 16), (CF_YEAST, 21, 4)])
def test_gpu_built_index_is_bit_identical_to_host_built(prefix, w, skew, tmp_path):
    """mz_sshash_from_unitig_set_device (minimizer scan, sort, records, EF, entries, skew MPHF on the GPU) must
    produce the very arrays of the host builder: compared through the flat index file, then queried."""
    us = M.UnitigSet.from_cf_prefix(prefix)
    host = M.SSHash.from_unitig_set(us, w, skew)
    dev = M.SSHash.from_unitig_set_on_device(us, w, skew)   # (yeast, 15, 16) exercises the transparent host fallback: direct records do not pay for so few minimizers
    assert not host.built_on_device()
    assert dev.built_on_device() == ((prefix, w) != (CF_YEAST, 15)), "the GPU builder handed a configuration it should build to the host"
    host.save(tmp_path / "h.mz")
    dev.save(tmp_path / "d.mz")
    a, b = open(tmp_path / "h.mz", "rb").read(), open(tmp_path / "d.mz", "rb").read()
    assert len(a) == len(b)
    if a != b:
        diff = next(i for i in range(len(a)) if a[i] != b[i])
        raise AssertionError(f"index files differ first at byte {diff} of {len(a)}")
    dev.validate_self()


def test_host_pointer_batch_walks_several_chunks(yeast, monkeypatch):
    """mz_sshash_k2u_batch pipelines chunks through its pinned slots (13 and 3 chunks: drained by the helper thread; 2 and 1: by the caller): odd chunk size, partial last chunk"""
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, 15, 16)
    ix.to_device(0)
    q = H.query_mix(words, starts, k, 50_001, seed=77)
    want, wc = orc.k2u_batch(q, n_threads=2)
    for chunk in (4097, 25_000, 50_001, 50_000):
        monkeypatch.setenv("MZ_HOST_CHUNK", str(chunk))
        got, counts = ix.k2u_batch(q)
        assert got.tobytes() == want.tobytes(), chunk
        assert list(counts)[:2] == [int((want["match"] != 0).sum()), int((want["match"] == 0).sum())], chunk
    monkeypatch.delenv("MZ_HOST_CHUNK")
    got, _ = ix.k2u_batch(q[:1])
    assert got.tobytes() == want[:1].tobytes()


def test_host_pointer_reads_walk_several_chunks(yeast, monkeypatch):
    """mz_sshash_k2u_reads pipelines runs of whole reads through the staging slots; a read too long for a chunk takes the
    unchunked path; results equal the oracle's walk either way"""
    k, seqs = yeast
    ix, orc, _ = both(seqs, k, 15, 16)
    ix.to_device(0)
    ref = H.read_fasta(CF_YEAST + ".fa")[0][1]
    rng = np.random.default_rng(5)
    reads = []
    for _ in range(400):
        a = int(rng.integers(0, len(ref) - 400))
        reads.append(ref[a:a + int(rng.integers(0, 300))])
    reads[7] = "ACGTN" * 20
    bases, ro = reads_fixture([("", r) for r in reads])
    want, ko, oc = orc.k2u_reads(bases, ro, streaming=False)
    for chunk in (2048, 5000, 1 << 21):
        monkeypatch.setenv("MZ_HOST_CHUNK", str(chunk))
        got, counts = ix.k2u_reads(bases, ro)
        assert got.tobytes() == want.tobytes() and list(counts) == oc.tolist(), chunk
    long_reads = [ref[1000:9000], ref[20:90], ref[50000:50100]]  # 7970 windows in one read > chunk of 2048: unchunked path
    bases, ro = reads_fixture([("", r) for r in long_reads])
    want, ko, oc = orc.k2u_reads(bases, ro, streaming=False)
    monkeypatch.setenv("MZ_HOST_CHUNK", "2048")
    got, counts = ix.k2u_reads(bases, ro)
    assert got.tobytes() == want.tobytes() and list(counts) == oc.tolist()


def test_concurrent_host_threads_on_one_handle(yeast):
    """kphf/mod.rs:105-109 requires Self: Sync: several host threads query one index concurrently (ctypes drops the GIL)"""
    import threading
    k, seqs = yeast
    ix, orc, (words, total, starts) = both(seqs, k, 15, 16)
    ix.to_device(0)
    qs = [H.query_mix(words, starts, k, 150000 + 1000 * i, seed=100 + i) for i in range(6)]
    wants = [orc.k2u_batch(q, n_threads=2)[0] for q in qs]
    gots = [None] * len(qs)

    def work(i):
        for _ in range(3):
            gots[i] = ix.k2u_batch(qs[i])[0]
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(qs))]
    [t.start() for t in th]
    [t.join() for t in th]
    for g, w_ in zip(gots, wants):
        assert g.tobytes() == w_.tobytes()
