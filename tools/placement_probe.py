#!/usr/bin/env python3
"""Does the physical placement of the index arrays matter?  One process, the same index uploaded several times
(separate handles from one saved file), the same queries timed on each replica."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mazu_amd as M
from mazu_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "s-human"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000_000
dev = torch.device("cuda", 0)
us = synth.make_unitig_set(wl, device=dev)
uset = M.UnitigSet.from_packed(us["words"].cpu().numpy().view(np.uint64), us["total_len"], us["starts"].cpu().numpy().view(np.uint64), 31)
ix0 = M.SSHash.from_unitig_set_on_device(uset, 15, 32)
path = "/dev/shm/placement_probe.mz"
ix0.save(path)
q = synth.make_query_mix(ix0, us, n, seed=1000, device=dev)
out = torch.empty(n * 2, dtype=torch.int64, device=dev)
counts = torch.zeros(3, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream(dev)
handles = [ix0] + [M.SSHash.load(path) for _ in range(3)]
for h in handles[1:]:
    h.to_device(0)
for rnd in range(2):
    for j, h in enumerate(handles):
        for _ in range(2):
            h.k2u_batch_device(q, out, counts, stream.cuda_stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(5):
            h.k2u_batch_device(q, out, counts, stream.cuda_stream)
        b.record(stream)
        torch.cuda.synchronize()
        print(f"round {rnd} replica {j}: {a.elapsed_time(b) / 5:.3f} ms per launch", flush=True)
os.remove(path)
