#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer reads entry point mz_sshash_k2u_reads."""
import sys, time, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mazu_amd as M
from mazu_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "s-txome"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
us = synth.make_unitig_set(wl, device="cuda")
uset = M.UnitigSet.from_packed(us["words"].cpu().numpy().view(np.uint64), us["total_len"], us["starts"].cpu().numpy().view(np.uint64), 31)
ix = M.SSHash.from_unitig_set_on_device(uset, 15, 32)
bases, ro = synth.make_reads(us, n_reads, 150, seed=3, device="cuda")
bases = bases.cpu().numpy()
ro = ro.cpu().numpy().view(np.uint64)
ix.k2u_reads(bases[:150 * 1000], ro[:1001])
best = 1e9
for _ in range(3):
    t0 = time.time()
    out, counts = ix.k2u_reads(bases, ro)
    best = min(best, time.time() - t0)
print(f"{wl}: host-pointer k2u_reads of {n_reads} x 150 bp = {len(out)} windows: {best*1e3:.1f} ms = {len(out)/best/1e6:.1f} M windows/s (PCIe-inclusive, best of 3, fresh result array per call), counts {list(counts)}", flush=True)
best = 1e9
for _ in range(3):
    t0 = time.time()
    ix.k2u_reads(bases, ro, out=out)
    best = min(best, time.time() - t0)
print(f"{wl}: same into a result array the caller reuses: {best*1e3:.1f} ms = {len(out)/best/1e6:.1f} M windows/s", flush=True)
