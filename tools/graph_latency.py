#!/usr/bin/env python3
"""Small device-resident batches: mz_sshash_k2u_batch_device + mz_u2pos_project_device_noalloc called eagerly against the same
pair recorded once in a HIP graph and replayed (runs ON the GPU box)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mazu_amd as M  # noqa: E402
from mazu_amd import synth  # noqa: E402
from mazu_amd._capi import check, lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "s-eighth"
us = synth.make_unitig_set(wl, device="cuda")
starts = us["starts"].cpu().numpy().view(np.uint64)
ix = M.SSHash.from_unitig_set_on_device(M.UnitigSet.from_packed(us["words"].cpu().numpy().view(np.uint64), us["total_len"], starts, 31), 15, 32)
n_u = us["n_unitigs"]
ctable = (np.uint64(1) << np.uint64(63)) | ((starts[:-1] & np.uint64((1 << 30) - 1)) << np.uint64(32)) | (starts[:-1] >> np.uint64(30))
offs_t = np.arange(n_u + 1, dtype=np.uint64)
u = C.c_void_p()
check(lib.mz_u2pos_dense_new(ctable.ctypes.data, n_u, offs_t.ctypes.data, n_u, C.byref(u)))
check(lib.mz_u2pos_to_device(u, 0))
for n in (1_000, 10_000, 100_000, 1_000_000):
    q = synth.make_query_mix(ix, us, n, seed=5, device="cuda")
    recs = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    out = torch.zeros(2 * n, dtype=torch.int64, device="cuda")

    def run(stream):
        ix.k2u_batch_device(q, recs, None, stream)
        check(lib.mz_u2pos_project_device_noalloc(u, 31, recs.data_ptr(), n, off.data_ptr(), out.data_ptr(), n, stream))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run(torch.cuda.current_stream().cuda_stream)
    reps = 300
    s = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run(s)
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / reps
    print(f"{wl}: {n} queries, lookup + projection: eager {t_eager * 1e6:.1f} us, graph replay {t_graph * 1e6:.1f} us per batch "
          f"({n / t_graph / 1e6:.1f} M k-mers/s)", flush=True)
lib.mz_u2pos_free(u)
