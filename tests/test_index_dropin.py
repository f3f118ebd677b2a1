"""The drop-in property (reference README.md:17-38; src/pf1/dense_index.rs:213-328): load a
pufferfish dense index, pull out its UnitigSet, put an SSHash in its place with
ModIndex::from_parts, and validate_self / as_streaming keep working.
CPU part = BASELINE config 0 (plumbing, K2U played by the oracle); GPU part = the real drop-in."""
import json
import os

import numpy as np
import pytest

import helpers as H
import mazu_amd as M
from mazu_amd import index as I
from mazu_amd import pf1
from oracle import oracle as O

KATS = json.load(open(os.path.join(H.GOLDEN, "reference_kats.json")))
PF1 = os.path.join(H.GOLDEN, "pf1")


class OracleK2U:
    """TEST-ONLY adapter: lets the CPU oracle play the K2U role in the host mirror (config 0)."""

    def __init__(self, words, total, starts, k, w, skew):
        self.o = O.OracleSSHash(np.concatenate([words, np.zeros(2, dtype=np.uint64)]), total, starts, k, w, skew, with_dict=False)

    def k(self): return self.o.k()
    def n_unitigs(self): return self.o.n_unitigs()
    def n_kmers(self): return self.o.n_kmers()
    def k2u_batch(self, fw): return self.o.k2u_batch(fw, n_threads=4)
    def k2u_reads(self, bases, ro, streaming=False): r, _, c = self.o.k2u_reads(bases, ro, streaming); return r, c


def gpu_factory(w, skew):
    def f(words, total, starts, k):
        return M.SSHash.from_unitig_set(M.UnitigSet.from_packed(words, total, starts, k), w, skew)
    return f


def test_pf1_fixture_decoding():  # dense_index.rs:117-129, cpp.rs:245-271
    f = pf1.DenseIndexFiles(os.path.join(PF1, "tiny_index"))
    assert f.n_unitigs == KATS["n_unitigs"]["tiny_index"] and f.k == 3 and f.total_len == 6
    assert f.pos_width == 3 and pf1.unpack(f.pos_words, 3, f.n_pos).tolist() == [3, 2, 0, 1]
    f = pf1.DenseIndexFiles(os.path.join(PF1, "yeast_chr01_index"))
    assert f.n_unitigs == KATS["n_unitigs"]["yeast_chr01_index"] and f.pos_width == 18
    assert f.total_len == 239228 and int(f.starts[-1]) == 239228


@pytest.mark.parametrize("name,w,skew", [("tiny_index", 2, O.NO_SKEW), ("tiny-rc_index", 2, O.NO_SKEW),
                                         ("tiny-multi-refs/tiny-multi-refs_index", 3, 0), ("small_txome_index", 15, 32),
                                         ("yeast_chr01_index", 15, 32)])
def test_config0_plumbing_on_cpu(name, w, skew):  # dense_index.rs:274-328 with the oracle as K2U
    idx, f = I.dense_index_from_pf1(os.path.join(PF1, name), lambda a, b, c, k: OracleK2U(a, b, c, k, w, skew))
    n = idx.validate_self()
    assert n == sum(max(0, idx.refs.ref_len(r) - f.k + 1) for r in range(idx.refs.n_refs()))
    if name == "yeast_chr01_index":
        assert n == 230188 and idx.n_unitigs() == 577


def test_tiny_k3_known_ref_positions_on_cpu():  # dense_index.rs:168-265
    idx, f = I.dense_index_from_pf1(os.path.join(PF1, "tiny_index"), lambda a, b, c, k: OracleK2U(a, b, c, k, 2, O.NO_SKEW))
    kat = KATS["tiny_k3_refpos"]
    for hit in kat["hits"]:
        km = M.CanonicalKmer(hit["kmer"])
        assert idx.get_ref_pos_eager(km) == [I.MappedRefPos.new_fw(hit["ref_id"], hit["pos"])]
        km.swap()
        assert idx.get_ref_pos_eager(km) == [I.MappedRefPos.new_rc(hit["ref_id"], hit["pos"])]
    for miss in kat["misses"]:
        assert idx.get_ref_pos_eager(M.CanonicalKmer(miss)) is None
    with pytest.raises(ValueError):  # dense_index.rs:148-163 (#[should_panic])
        idx.get_ref_pos_eager(M.CanonicalKmer("aaaaaa"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,skew", [("tiny_index", 2, M.MZ_NO_SKEW_INDEX), ("tiny-multi-refs/tiny-multi-refs_index", 3, 0),
                                         ("small_txome_index", 15, 32), ("yeast_chr01_index", 15, 32), ("yeast_chr01_index", 15, 8)])
def test_readme_dropin_on_gpu(name, w, skew):  # README.md:17-38, dense_index.rs:314-328
    idx, f = I.dense_index_from_pf1(os.path.join(PF1, name), gpu_factory(w, skew))
    idx.as_k2u().validate_self()                      # K2U::validate_self on the device
    n = idx.validate_self()                           # Validate::validate_self over the stored references
    if name == "yeast_chr01_index":
        assert n == 230188
    # .as_streaming(): walk every stored reference through the reads kernel (caching.rs:175-218)
    st = idx.as_streaming()
    for r in range(idx.refs.n_refs()):
        fw = idx.refs.kmer_words(r, f.k)
        seq = "".join(H.decode_kmer(int(x), 1) for x in (fw & np.uint64(3))) if len(fw) else ""
        if len(fw):
            seq += H.decode_kmer(int(fw[-1]), f.k)[1:]
            assert st.validate_ckmers(r, seq) == len(fw)


@pytest.mark.gpu
def test_tiny_k3_known_ref_positions_on_gpu():  # dense_index.rs:213-265 (SSHash w=2 dropped into tiny_index)
    idx, f = I.dense_index_from_pf1(os.path.join(PF1, "tiny_index"), gpu_factory(2, M.MZ_NO_SKEW_INDEX))
    for hit in KATS["tiny_k3_refpos"]["hits"]:
        km = M.CanonicalKmer(hit["kmer"])
        assert idx.get_ref_pos_eager(km) == [I.MappedRefPos.new_fw(hit["ref_id"], hit["pos"])]
        km.swap()
        assert idx.get_ref_pos_eager(km) == [I.MappedRefPos.new_rc(hit["ref_id"], hit["pos"])]
    for miss in KATS["tiny_k3_refpos"]["misses"]:
        assert idx.get_ref_pos_eager(M.CanonicalKmer(miss)) is None


@pytest.mark.gpu
def test_streaming_validate_fasta_piscem_style():  # caching.rs:238-253 / piscem_index.rs:63-99 on cf fixtures
    """validate_fasta through .as_streaming(): every ACGT window of the FASTA the unitigs were built
    from must be found (tiny.fa contains Ns)."""
    for prefix, w, skew in ((os.path.join(H.GOLDEN, "cf", "tiny", "tiny"), 3, M.MZ_NO_SKEW_INDEX),
                            (os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7"), 15, 16)):
        ix = M.SSHash.from_unitig_set(M.UnitigSet.from_cf_prefix(prefix), w, skew)
        for _, seq in H.read_fasta(prefix + ".fa"):
            bases = np.frombuffer(seq.encode(), dtype=np.uint8)
            recs, counts = ix.k2u_reads(bases, np.array([0, len(bases)], dtype=np.uint64), streaming=True)
            assert counts[1] == 0 and counts[0] + counts[2] == len(recs) and counts[0] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,skew", [("tiny-multi-refs/tiny-multi-refs_index", 3, 0), ("small_txome_index", 15, 32), ("yeast_chr01_index", 15, 32)])
def test_projection_on_device_matches_host_projection(name, w, skew, monkeypatch):  # index.rs:184-216, dense_unitig_table.rs:55-76
    idx, f = I.dense_index_from_pf1(os.path.join(PF1, name), gpu_factory(w, skew))
    n_dev = idx.validate_self_on_device()               # lookup + projection + assertion on the GPU
    assert n_dev == idx.validate_self()                 # host projection of the same GPU lookups
    # record-level parity of get_ref_pos_batch with the host mirror of project_onto_u_occs, incl. misses
    rng = np.random.default_rng(4)
    fw = np.concatenate([idx.refs.kmer_words(r, f.k)[:: max(1, idx.refs.ref_len(r) // 4000)] for r in range(idx.refs.n_refs())])
    fw = np.concatenate([fw, H.rc_words(fw[::3], f.k), rng.integers(0, 1 << (2 * f.k), 500, dtype=np.uint64)])
    offs, recs = idx.get_ref_pos_batch(fw, cap=8)       # tiny cap: exercises the grow-and-retry path
    k2u_recs, _ = idx.k2u.k2u_batch(fw)
    assert len(offs) == len(fw) + 1 and int(offs[-1]) == len(recs)
    for i in rng.integers(0, len(fw), 300):
        got = recs[int(offs[i]): int(offs[i + 1])]
        if int(k2u_recs[i]["match"]) == 0:
            assert len(got) == 0
            continue
        ref_id, cpos, fwd = I.DenseUnitigTable.decode_pf1(idx.u2pos.encoded_unitig_occs(int(k2u_recs[i]["unitig_id"])))
        want = I.project_onto_u_occs(f.k, k2u_recs[i], ref_id, cpos, fwd)
        assert [(int(g["ref_id"]), int(g["pos"]), int(g["o"])) for g in got] == [(m.ref_id, m.pos, m.o) for m in want]
        assert (got["query"] == i).all()
    assert idx.u2pos.num_total_occs() == len(f.ctable)
    # the batch walked in chunks through the staging slots (odd sizes, a partial last chunk, record buffers that have to grow,
    # a caller's array that ends inside a chunk): same offsets, same records, query indices of the whole batch
    for chunk, cap in (("1000", None), ("77", 8), ("4096", len(recs) // 2 + 1)):
        monkeypatch.setenv("MZ_REFPOS_CHUNK", chunk)
        if cap is None or cap == 8:
            o2, r2 = idx.get_ref_pos_batch(fw, cap=cap)
            assert o2.tobytes() == offs.tobytes() and r2.tobytes() == recs.tobytes(), chunk
        else:  # one call into an array that is too small: the records that fit are there, the total is reported
            import ctypes as C
            from mazu_amd._capi import check, lib
            o2 = np.zeros(len(fw) + 1, dtype=np.uint64)
            r2 = np.zeros(cap, dtype=I.MAPPED_DTYPE)
            n_out = C.c_uint64(0)
            check(lib.mz_get_ref_pos_batch(idx.k2u._h, idx.u2pos.handle(), fw.ctypes.data, len(fw), o2.ctypes.data, r2.ctypes.data, cap, C.byref(n_out)))
            assert n_out.value == len(recs) and o2.tobytes() == offs.tobytes() and r2.tobytes() == recs[:cap].tobytes()


def test_spt_tiny_known_answers():  # src/spt.rs:156-211
    spt = I.SPT.from_cf_prefix(os.path.join(H.GOLDEN, "cf", "tiny", "tiny"))
    assert spt.num_refs() == 2 and spt.n_total_occs() == 4
    assert spt.ref_name(0) == "Reference:1_Sequence:I" and spt.ref_len(0) == 3 + 10 + 1 + 10
    assert spt.get_occ(0, 0) == (0, 3, True)
    assert spt.get_occ(0, 1) == (1, 10 + 1, False)
    assert spt.get_occ(1, 0) == (0, 3 + 10 + 1, False)
    assert spt.get_occ(1, 1) == (1, 0, True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,skew", [("tiny", 3, M.MZ_NO_SKEW_INDEX), ("yeast_chr7", 15, 16)])
def test_index_from_cuttlefish_prefix_validate_fasta(name, w, skew):  # defaults.rs:60-70, piscem_index.rs:63-99
    prefix = os.path.join(H.GOLDEN, "cf", name, name)
    idx, spt = I.index_from_cf_prefix(prefix, lambda seqs, k: M.SSHash.from_unitig_set_on_device(M.UnitigSet.from_seqs(seqs, k), w, skew))
    recs = H.read_fasta(prefix + ".fa")
    assert [spt.ref_len(i) for i in range(spt.num_refs())] == [len(s) for _, s in recs]
    n_dev = I.validate_fasta_on_device(idx, prefix + ".fa")       # reads kernel + projection + assertion on the GPU
    st = idx.as_streaming()                                       # .as_streaming().validate_fasta with the host projection
    n_host = sum(st.validate_ckmers(ri, seq) for ri, (_, seq) in enumerate(recs))
    assert n_dev == n_host and n_dev > 0


def test_piscem_occurrence_codec_known_answers():  # spt_compact.rs:504-520 (occ_encoding), :398-404 (masks), :407-430 (compare_spt_impls)
    t = I.PiscemUnitigTable.from_occs([0], [1], [False], [0, 1], ["r"], max_ref_len=2)   # ref_shift 3 -> word 0b010
    assert (t.ref_shift, t.pos_mask) == (3, 0b11)
    assert int(I._unpack(t.ctable_words, [0], t.occ_width)[0]) == 0b010
    r, p_, f = t.decode_occs(np.array([0]))
    assert (int(r[0]), int(p_[0]), bool(f[0])) == (0, 1, False)
    for i in range(1, 64):  # POS_MASKS[i] has bit width i
        assert I._bit_width((1 << i) - 1) == i
    for name in ("tiny", "yeast_chr7"):  # the compact table decodes to the very occurrences of the 64-bit table
        spt = I.SPT.from_cf_prefix(os.path.join(H.GOLDEN, "cf", name, name))
        rid, pos, fwd = I.DenseUnitigTable.decode_pf1(spt.ctable)
        t = I.PiscemUnitigTable.from_occs(rid, pos, fwd, spt.offsets, spt.ref_names, int(max(spt.ref_lens)))
        assert t.occ_width == 1 + int(max(spt.ref_lens)).bit_length() + len(spt.ref_names).bit_length() and t.occ_width < 64
        r2, p2, f2 = t.decode_occs(np.arange(len(spt.ctable)))
        assert np.array_equal(r2, rid) and np.array_equal(p2, pos) and np.array_equal(f2, fwd)
        assert np.array_equal(I._unpack(t.contig_offsets_words, np.arange(len(spt.offsets)), t.offsets_width), spt.offsets)
    spt = I.SPT.from_cf_prefix(os.path.join(H.GOLDEN, "cf", "tiny", "tiny"))   # spt_compact.rs:432-500 through the compact table
    rid, pos, fwd = I.DenseUnitigTable.decode_pf1(spt.ctable)
    t = I.PiscemUnitigTable.from_occs(rid, pos, fwd, spt.offsets, spt.ref_names, int(max(spt.ref_lens)))
    got = [tuple(int(x[0]) if j < 2 else bool(x[0]) for j, x in enumerate(t.decode_occs(np.array([i])))) for i in range(4)]
    assert got == [(0, 3, True), (1, 11, False), (0, 14, False), (1, 0, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,skew", [("tiny", 3, 2), ("yeast_chr7", 15, 32)])
def test_piscem_index_validate_fasta_and_projection_parity(name, w, skew):  # piscem_index.rs:63-99, dense_unitig_table.rs:127-153
    """PiscemIndex::from_cf_prefix with the occurrences packed into the minimal number of bits: validate_fasta with lookup,
    projection and assertion on the GPU (plain and through .as_streaming()), and record-for-record the same MappedRefPos
    lists as the 64-bit DenseUnitigTable"""
    prefix = os.path.join(H.GOLDEN, "cf", name, name)
    factory = lambda seqs, k: M.SSHash.from_unitig_set_on_device(M.UnitigSet.from_seqs(seqs, k), w, skew)  # noqa: E731
    pidx, spt = I.index_from_cf_prefix(prefix, factory, piscem=True)
    didx, _ = I.index_from_cf_prefix(prefix, factory)
    assert pidx.u2pos.occ_width < 64 and pidx.u2pos.num_total_occs() == spt.n_total_occs() == didx.u2pos.num_total_occs()
    assert [pidx.u2pos.n_unitig_occs(u) for u in range(min(50, spt.n_unitigs()))] == [didx.u2pos.n_unitig_occs(u) for u in range(min(50, spt.n_unitigs()))]
    recs = H.read_fasta(prefix + ".fa")
    n_dev = I.validate_fasta_on_device(pidx, prefix + ".fa")
    st = pidx.as_streaming()
    assert n_dev == sum(st.validate_ckmers(ri, seq) for ri, (_, seq) in enumerate(recs)) and n_dev > 0
    rng = np.random.default_rng(3)
    words, total, starts = H.pack_seqs(spt.seqs)
    fw = np.concatenate([H.query_mix(words, starts, spt.k(), 20_000, seed=8), rng.integers(0, 1 << (2 * spt.k()), 500, dtype=np.uint64)])
    po, pr = pidx.get_ref_pos_batch(fw)
    do, dr = didx.get_ref_pos_batch(fw)
    assert po.tobytes() == do.tobytes() and pr.tobytes() == dr.tobytes() and len(pr) > 5_000


@pytest.mark.gpu
def test_get_ref_pos_batch_grows_its_record_buffers(monkeypatch):
    """unitigs with 40 occurrences each: a chunk's records outgrow the two-per-query buffers its staging slot starts with, so the slot
    grows and the fill is repeated; chunked and unchunked calls agree, every query has 40 records or none, and sampled records equal
    the host projection (index.rs:184-216)"""
    k, seqs = H.load_cf_seqs(os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7"))
    words, total, starts = H.pack_seqs(seqs)
    n_u, per = len(seqs), 40
    j = np.tile(np.arange(per, dtype=np.uint64), n_u)
    u = np.repeat(np.arange(n_u, dtype=np.uint64), per)
    fwd = (u + j) & np.uint64(1)
    ctable = (fwd << np.uint64(63)) | ((np.uint64(5000) * j + u) << np.uint64(32)) | j   # [63] forward, [62:32] position, [31:0] reference
    table = I.DenseUnitigTable(ctable, np.arange(n_u + 1, dtype=np.uint64) * np.uint64(per), [f"r{r}" for r in range(per)])
    k2u = M.SSHash.from_unitig_set(M.UnitigSet.from_packed(words, total, starts, k), 15, 16)
    idx = I.ModIndex.from_parts(None, k2u, table, None)
    fw = H.query_mix(words, starts, k, 6000, seed=21)
    recs_k2u, _ = k2u.k2u_batch(fw)
    n_hit = int((recs_k2u["match"] != 0).sum())
    offs, recs = idx.get_ref_pos_batch(fw)                       # one chunk: 6 000 queries, ~120 000 records against room for 16 384
    assert int(offs[-1]) == len(recs) == per * n_hit and n_hit > 2000
    assert set(np.diff(offs.astype(np.int64)).tolist()) == {0, per}
    for chunk in ("1000", "4097"):
        monkeypatch.setenv("MZ_REFPOS_CHUNK", chunk)             # ~20 000 records per chunk against room for 4 096
        o2, r2 = idx.get_ref_pos_batch(fw)
        assert o2.tobytes() == offs.tobytes() and r2.tobytes() == recs.tobytes(), chunk
    # counting only (no record array): the offsets and the total, nothing else
    import ctypes as C
    from mazu_amd._capi import check, lib
    o3 = np.zeros(len(fw) + 1, dtype=np.uint64)
    n_out = C.c_uint64(0)
    check(lib.mz_get_ref_pos_batch(k2u._h, table.handle(), fw.ctypes.data, len(fw), o3.ctypes.data, None, 0, C.byref(n_out)))
    assert n_out.value == len(recs) and o3.tobytes() == offs.tobytes()
    rng = np.random.default_rng(2)
    for i in rng.choice(np.flatnonzero(recs_k2u["match"] != 0), 50, replace=False):
        got = recs[int(offs[i]): int(offs[i + 1])]
        ref_id, cpos, f = I.DenseUnitigTable.decode_pf1(table.encoded_unitig_occs(int(recs_k2u[i]["unitig_id"])))
        want = I.project_onto_u_occs(k, recs_k2u[i], ref_id, cpos, f)
        assert [(int(g["ref_id"]), int(g["pos"]), int(g["o"])) for g in got] == [(m.ref_id, m.pos, m.o) for m in want]
        assert (got["query"] == i).all()
