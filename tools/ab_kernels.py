#!/usr/bin/env python3
"""Same-process A/B of kernel variants (runs ON the GPU box).

    python tools/ab_kernels.py --workload s-human --steps 10 base=mazu_amd/libmazu_amd.so v1=build/ab/v1.so ...

The index is built ONCE with the in-tree library; every variant library (same host structures, different mz_device.hip
build) then runs the same device-resident batch through ITS mz_sshash_k2u_batch_device on that handle.  Variants are
timed interleaved (`--reps` rounds) with events on the launch stream, and every variant's output is compared with the
first one's (variants built with MZ_STOP_AFTER are expected to differ: pass --no-compare)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mazu_amd as M  # noqa: E402
from mazu_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="s-human")
ap.add_argument("--batch", type=int, default=125_000_000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--neg-frac", type=float, default=0.5)
ap.add_argument("--neg-kind", default="uniform")
ap.add_argument("--no-compare", action="store_true")
ap.add_argument("--reads", type=int, default=0, help="time mz_sshash_k2u_reads_device on this many 150-bp reads instead of point queries")
ap.add_argument("variants", nargs="+", help="name=path/to/lib.so")
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
us = synth.make_unitig_set(args.workload, device=dev)
words = us["words"].cpu().numpy().view(np.uint64)
starts = us["starts"].cpu().numpy().view(np.uint64)
ix = M.SSHash.from_unitig_set_on_device(M.UnitigSet.from_packed(words, us["total_len"], starts, 31), 15, 32)
ix.to_device(0)
print(f"index: {ix.device_bytes() / 1e9:.2f} GB, lines={ix.has_minimizer_lines()}", flush=True)
stream = torch.cuda.current_stream(dev)
if args.reads:
    bases, ro = synth.make_reads(us, args.reads, 150, seed=2000, device=dev)
    ko = synth.kmer_offsets(ro, 31)
    args.batch = int(ko[-1])
else:
    q = synth.make_query_mix(ix, us, args.batch, seed=1000, device=dev, neg_frac=args.neg_frac, neg_kind=args.neg_kind)
libs = {}
for v in args.variants:
    name, path = v.split("=", 1)
    L = C.CDLL(os.path.abspath(path))
    L.mz_sshash_k2u_batch_device.restype = C.c_int
    L.mz_sshash_k2u_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mz_sshash_k2u_reads_device.restype = C.c_int
    L.mz_sshash_k2u_reads_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    libs[name] = L
outs = {n: torch.zeros(args.batch * 2, dtype=torch.int64, device=dev) for n in libs} if not args.no_compare else {}
scratch = torch.zeros(args.batch * 2, dtype=torch.int64, device=dev)
cnt = torch.zeros(3, dtype=torch.int64, device=dev)


def run(L, out):
    if args.reads:
        rc = L.mz_sshash_k2u_reads_device(ix._h, bases.data_ptr(), bases.numel(), ro.data_ptr(), ko.data_ptr(), ro.numel() - 1, 1, out.data_ptr(),
                                          cnt.data_ptr(), stream.cuda_stream)
    else:
        rc = L.mz_sshash_k2u_batch_device(ix._h, q.data_ptr(), q.numel(), out.data_ptr(), cnt.data_ptr(), stream.cuda_stream)
    assert rc == 0, rc


res = {n: [] for n in libs}
for n, L in libs.items():  # warm-up (and the outputs that are compared)
    run(L, outs.get(n, scratch))
torch.cuda.synchronize()
for rep in range(args.reps):
    for n, L in libs.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(args.steps):
            run(L, scratch)
        b.record(stream)
        torch.cuda.synchronize()
        res[n].append(a.elapsed_time(b) / args.steps)
base = next(iter(libs))
for n in libs:
    ms = min(res[n])
    same = None if args.no_compare else bool(torch.equal(outs[n], outs[base]))
    print(json.dumps({"variant": n, "ms": [round(x, 4) for x in res[n]], "G_kmers_per_s": round(args.batch / ms / 1e6, 2), "identical_to_" + base: same}), flush=True)
