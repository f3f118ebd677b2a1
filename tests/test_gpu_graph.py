"""Lookup + projection captured once in a HIP graph and replayed: the device-pointer entry points launch kernels only (no
allocation, no memset node, no synchronisation), so a serving loop can record `mz_sshash_k2u_batch_device` followed by
`mz_u2pos_project_device_noalloc` and replay the pair on new queries.  Every replay -- with eager work and synchronisations of
the program's own between the replays -- must give what the eager calls give on the same queries."""
import os

import numpy as np
import pytest

import helpers as H
import mazu_amd as M
from mazu_amd import index as I
from mazu_amd._capi import check, lib

pytestmark = pytest.mark.gpu

PF1 = os.path.join(H.GOLDEN, "pf1")


def test_lookup_and_projection_replay_in_a_graph():
    import torch
    idx, f = I.dense_index_from_pf1(os.path.join(PF1, "yeast_chr01_index"), lambda words, total, starts, k:
                                    M.SSHash.from_unitig_set_on_device(M.UnitigSet.from_packed(words, total, starts, k), 15, 32))
    k, u = f.k, idx.u2pos.handle()
    dev = torch.cuda.current_device()
    check(lib.mz_u2pos_to_device(u, dev))
    members = idx.refs.kmer_words(0, k)
    n, cap = 40_000, 400_000
    rng = np.random.default_rng(12)

    def batch():
        fw = np.concatenate([rng.choice(members, n // 2), H.rc_words(rng.choice(members, n // 4), k),
                             rng.integers(0, 1 << (2 * k), n - n // 2 - n // 4, dtype=np.uint64)])
        rng.shuffle(fw)
        return torch.from_numpy(fw.view(np.int64)).cuda()

    d_q = batch()
    d_recs = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    d_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    d_out = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")

    def run(q, recs, off, out, stream):
        idx.k2u.k2u_batch_device(q, recs, None, stream)
        check(lib.mz_u2pos_project_device_noalloc(u, k, recs.data_ptr(), n, off.data_ptr(), out.data_ptr(), cap, stream))

    side = torch.cuda.Stream()  # warm-up on a side stream, as graph capture wants it
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(d_q, d_recs, d_off, d_out, side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run(d_q, d_recs, d_off, d_out, torch.cuda.current_stream().cuda_stream)

    e_recs, e_off, e_out = torch.zeros_like(d_recs), torch.zeros_like(d_off), torch.zeros_like(d_out)
    totals = []
    for rep in range(4):
        d_q.copy_(batch())                       # new queries into the buffer the graph reads
        d_recs.fill_(-1); d_off.fill_(-1)        # the replay has to rewrite everything it reports
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        run(d_q, e_recs, e_off, e_out, torch.cuda.current_stream().cuda_stream)  # eager, same queries
        torch.cuda.synchronize()
        total = int(e_off[n].item())
        assert 0 < total <= cap
        assert torch.equal(d_recs, e_recs) and torch.equal(d_off, e_off) and torch.equal(d_out[: 2 * total], e_out[: 2 * total]), rep
        totals.append(total)
        _ = (e_out[: 2 * total].sum() + e_off.sum()).item()  # eager work of the program's own between the replays
    assert len(set(totals)) > 1  # the batches differed: the replays did not repeat one answer


def test_reads_lookup_replays_in_a_graph():
    """the reads entry point is a single kernel launch as well: captured once, replayed on new bases in the same buffers"""
    import torch
    CF = os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7")
    k, seqs = H.load_cf_seqs(CF)
    ix = M.SSHash.from_unitig_set_on_device(M.UnitigSet.from_seqs(seqs, k), 15, 16)
    ref = np.frombuffer(H.read_fasta(CF + ".fa")[0][1].encode(), dtype=np.uint8)
    n_reads, L = 2000, 150
    rng = np.random.default_rng(3)

    def reads():
        starts = rng.integers(0, len(ref) - L, n_reads)
        b = np.stack([ref[s:s + L] for s in starts])
        b[rng.random(b.shape) < 0.004] = ord("N")  # some skipped windows
        return torch.from_numpy(b.reshape(-1).copy()).cuda()

    ro = torch.arange(n_reads + 1, dtype=torch.int64, device="cuda") * L
    ko = torch.arange(n_reads + 1, dtype=torch.int64, device="cuda") * (L - k + 1)
    n_win = n_reads * (L - k + 1)
    d_b = reads()
    d_out = torch.zeros(2 * n_win, dtype=torch.int64, device="cuda")
    e_out = torch.zeros_like(d_out)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ix.k2u_reads_device(d_b, ro, ko, d_out, None, streaming=True, stream=side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ix.k2u_reads_device(d_b, ro, ko, d_out, None, streaming=True, stream=torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        d_b.copy_(reads())
        d_out.fill_(-1)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ix.k2u_reads_device(d_b, ro, ko, e_out, None, streaming=True, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(d_out, e_out), rep
        match = (e_out.view(torch.int32).view(-1, 4)[:, 3]).to(torch.int64)
        assert int((match == 1).sum() + (match == 2).sum()) > n_win // 2 and int((match == 3).sum()) > 0
