#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point mz_sshash_k2u_batch (H2D queries, kernel, D2H results)."""
import sys, time, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mazu_amd as M
from mazu_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "s-eighth"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
us = synth.make_unitig_set(wl, device="cuda")
ix = M.SSHash.from_unitig_set(M.UnitigSet.from_packed(us["words"].cpu().numpy().view(np.uint64), us["total_len"],
                                                       us["starts"].cpu().numpy().view(np.uint64), 31), 15, 32)
ix.to_device(0)
q = synth.make_query_mix(ix, us, n, seed=5, device="cuda").cpu().numpy().view(np.uint64)
ix.k2u_batch(q[:1_000_000])
for chunk in [None] + [int(c) for c in os.environ.get("CHUNKS", "").split(",") if c]:
    if chunk:
        os.environ["MZ_HOST_CHUNK"] = str(chunk)
        ix.k2u_batch(q[:4 * chunk])  # the pipe's buffers grow outside the timed calls
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        out, counts = ix.k2u_batch(q)
        best = min(best, time.time() - t0)
    print(f"{wl}: chunk {chunk or 'default'}: host-pointer k2u_batch of {n} queries: {best*1e3:.1f} ms = {n/best/1e6:.1f} M k-mers/s (PCIe-inclusive, best of 3, fresh result array per call), hits {counts[0]}", flush=True)
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        _, counts = ix.k2u_batch(q, out=out)
        best = min(best, time.time() - t0)
    print(f"{wl}: chunk {chunk or 'default'}: same into a result array the caller reuses: {best*1e3:.1f} ms = {n/best/1e6:.1f} M k-mers/s", flush=True)
os.environ.pop("MZ_HOST_CHUNK", None)
for m in (1, 1000, 100_000):
    t0 = time.time()
    reps = 200
    for _ in range(reps):
        ix.k2u_batch(q[:m])
    dt = (time.time() - t0) / reps
    print(f"  batch of {m}: {dt*1e6:.0f} us per call")
