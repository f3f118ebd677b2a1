"""mz_u2pos_project_device_noalloc -- the one entry point that allocates nothing -- against mz_u2pos_project_device.

Both are called through the C ABI on the same K2UPos records; offsets and MappedRefPos records must be byte-identical,
on two calls in a row, the second with a smaller input (the fixed scratch array keeps the first call's contents)."""
import os

import numpy as np
import pytest

import helpers as H
import mazu_amd as M
from mazu_amd import index as I
from mazu_amd._capi import MZ_EINVAL, MazuError, check, lib

pytestmark = pytest.mark.gpu

PF1 = os.path.join(H.GOLDEN, "pf1")
MAPPED = np.dtype([("ref_id", "<u4"), ("pos", "<u4"), ("o", "<u4"), ("query", "<u4")])


def _project(fn, u, k, d_recs, n, torch):
    """count pass, then fill pass, through entry point `fn`; returns (offsets, records) as numpy"""
    stream = torch.cuda.current_stream().cuda_stream
    d_off = torch.full((n + 1,), -1, dtype=torch.int64, device="cuda")
    check(fn(u, k, d_recs.data_ptr(), n, d_off.data_ptr(), None, 0, stream))
    torch.cuda.synchronize()
    total = int(d_off[n].item())
    d_out = torch.full((max(total, 1) * 2,), -1, dtype=torch.int64, device="cuda")
    d_off.fill_(-1)
    check(fn(u, k, d_recs.data_ptr(), n, d_off.data_ptr(), d_out.data_ptr(), total, stream))
    torch.cuda.synchronize()
    return d_off.cpu().numpy(), d_out[: 2 * total].cpu().numpy().view(MAPPED)


def test_noalloc_projection_equals_the_allocating_entry_point():
    import torch
    assert M.lib.mz_device_count() >= 1
    idx, f = I.dense_index_from_pf1(os.path.join(PF1, "yeast_chr01_index"), lambda words, total, starts, k:
                                    M.SSHash.from_unitig_set(M.UnitigSet.from_packed(words, total, starts, k), 15, 32))
    k = f.k
    u = idx.u2pos.handle()
    rng = np.random.default_rng(9)
    fw = idx.refs.kmer_words(0, k)
    fw = np.concatenate([fw, H.rc_words(fw[::2], k), rng.integers(0, 1 << (2 * k), 20_000, dtype=np.uint64)])
    rng.shuffle(fw)
    dev = torch.cuda.current_device()
    dq = torch.from_numpy(fw.view(np.int64)).cuda()
    d_recs = torch.empty(2 * len(fw), dtype=torch.int64, device="cuda")
    idx.k2u.k2u_batch_device(dq, d_recs, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()

    # before the set-up call the table is not resident: a clear error, no allocation behind the caller's back
    fresh = I.DenseUnitigTable(idx.u2pos.ctable, idx.u2pos.contig_offsets, idx.u2pos.ref_names)
    d_off = torch.zeros(len(fw) + 1, dtype=torch.int64, device="cuda")
    with pytest.raises(MazuError) as e:
        check(lib.mz_u2pos_project_device_noalloc(fresh.handle(), k, d_recs.data_ptr(), len(fw), d_off.data_ptr(), None, 0, None))
    assert e.value.code == MZ_EINVAL and "mz_u2pos_to_device" in str(e.value)

    check(lib.mz_u2pos_to_device(u, dev))
    for n in (len(fw), 12_345, 1, 0):  # the second and later calls are smaller than the first
        want_off, want = _project(lib.mz_u2pos_project_device, u, k, d_recs, n, torch)
        got_off, got = _project(lib.mz_u2pos_project_device_noalloc, u, k, d_recs, n, torch)
        assert got_off.tobytes() == want_off.tobytes(), n
        assert got.tobytes() == want.tobytes(), n
        if n == len(fw):
            assert len(want) > n // 2  # a real workload: most queries are reference k-mers with at least one occurrence
    # sizes beyond the fixed scratch array are refused
    cap = lib.mz_u2pos_noalloc_max_queries()
    assert cap == 1 << 24
    with pytest.raises(MazuError) as e:
        check(lib.mz_u2pos_project_device_noalloc(u, k, d_recs.data_ptr(), cap + 1, d_off.data_ptr(), None, 0, None))
    assert e.value.code == MZ_EINVAL and "exceed" in str(e.value)


def test_noalloc_after_the_table_became_resident_through_another_entry_point_and_across_streams():
    """the set-up call must resolve the scratch array of ITS device also when the table is resident already (it then returns early from the
    upload and must not rely on whatever device is current), and calls issued on two different streams back to back are ordered on
    the device by the library: both see correct results"""
    import torch
    idx, f = I.dense_index_from_pf1(os.path.join(PF1, "yeast_chr01_index"), lambda words, total, starts, k:
                                    M.SSHash.from_unitig_set(M.UnitigSet.from_packed(words, total, starts, k), 15, 32))
    k = f.k
    table = I.DenseUnitigTable(idx.u2pos.ctable, idx.u2pos.contig_offsets, idx.u2pos.ref_names)
    u = table.handle()
    fw = idx.refs.kmer_words(0, k)
    dq = torch.from_numpy(fw.view(np.int64)).cuda()
    n = len(fw)
    d_recs = torch.empty(2 * n, dtype=torch.int64, device="cuda")
    idx.k2u.k2u_batch_device(dq, d_recs, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want_off, want = _project(lib.mz_u2pos_project_device, u, k, d_recs, n, torch)  # makes the table resident
    check(lib.mz_u2pos_to_device(u, torch.cuda.current_device()))                    # early-return path of the upload
    total = len(want)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    offs = [torch.full((n + 1,), -1, dtype=torch.int64, device="cuda") for _ in range(2)]
    outs = [torch.full((max(total, 1) * 2,), -1, dtype=torch.int64, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for j, st in enumerate((s1, s2)):  # no synchronisation between the two calls
        check(lib.mz_u2pos_project_device_noalloc(u, k, d_recs.data_ptr(), n, offs[j].data_ptr(), outs[j].data_ptr(), total, st.cuda_stream))
    torch.cuda.synchronize()
    for j in range(2):
        assert offs[j].cpu().numpy().tobytes() == want_off.tobytes(), j
        assert outs[j][: 2 * total].cpu().numpy().view(MAPPED).tobytes() == want.tobytes(), j
