#!/usr/bin/env python3
"""PCIe-inclusive rate of mz_get_ref_pos_batch (host k-mers in, host MappedRefPos records out: lookup + projection on the GPU)."""
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
from mazu_amd.index import MAPPED_DTYPE  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "s-eighth"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
us = synth.make_unitig_set(wl, device="cuda")
starts = us["starts"].cpu().numpy().view(np.uint64)
ix = M.SSHash.from_unitig_set_on_device(M.UnitigSet.from_packed(us["words"].cpu().numpy().view(np.uint64), us["total_len"], starts, 31), 15, 32)
n_u = us["n_unitigs"]
# DenseUnitigTable in which every unitig occurs once, forward, on references of 2^30 bases cut from the concatenation (index.rs:315-346)
ctable = (np.uint64(1) << np.uint64(63)) | ((starts[:-1] & np.uint64((1 << 30) - 1)) << np.uint64(32)) | (starts[:-1] >> np.uint64(30))
offs_t = np.arange(n_u + 1, dtype=np.uint64)
u = C.c_void_p()
check(lib.mz_u2pos_dense_new(ctable.ctypes.data, n_u, offs_t.ctypes.data, n_u, C.byref(u)))
q = synth.make_query_mix(ix, us, n, seed=5, device="cuda").cpu().numpy().view(np.uint64)
offs = np.zeros(n + 1, dtype=np.uint64)
out = np.zeros(n, dtype=MAPPED_DTYPE)
n_out = C.c_uint64(0)


def call(m):
    check(lib.mz_get_ref_pos_batch(ix._h, u, q.ctypes.data, m, offs.ctypes.data, out.ctypes.data, n, C.byref(n_out)))


call(min(n, 4_000_000))
best = 1e9
for _ in range(3):
    t0 = time.time()
    call(n)
    best = min(best, time.time() - t0)
print(f"{wl}: mz_get_ref_pos_batch of {n} k-mers -> {n_out.value} records: {best*1e3:.1f} ms = {n/best/1e6:.1f} M k-mers/s (PCIe-inclusive, best of 3, "
      f"caller's arrays reused)", flush=True)
for m in (1, 1000):
    t0 = time.time()
    for _ in range(200):
        call(m)
    print(f"  batch of {m}: {(time.time() - t0) / 200 * 1e6:.0f} us per call")
lib.mz_u2pos_free(u)
