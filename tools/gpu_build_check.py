#!/usr/bin/env python3
"""host-built vs GPU-built index at scale: timings, file equality (cmp), validate_self."""
import os, sys, time, subprocess
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MZ_BUILD_TIMING"] = "1"
import mazu_amd as M
from mazu_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "s-eighth"
compare = len(sys.argv) > 2 and sys.argv[2] == "cmp"
us = synth.make_unitig_set(wl, device="cuda")
uset = M.UnitigSet.from_packed(us["words"].cpu().numpy().view(np.uint64), us["total_len"], us["starts"].cpu().numpy().view(np.uint64), 31)
t0 = time.time(); dev = M.SSHash.from_unitig_set_on_device(uset, 15, 32); t_dev = time.time() - t0
print(f"{wl}: GPU build {t_dev:.2f} s; {dev.n_minimizers()} minimizers, {dev.n_minimizer_occs()} occs, {dev.n_kmers_in_skew_index()} skew k-mers", flush=True)
rep = dev.validate_self_report(); print("validate_self", rep, flush=True)
assert rep["n_fail"] == 0
if compare:
    t0 = time.time(); host = M.SSHash.from_unitig_set(uset, 15, 32); t_host = time.time() - t0
    print(f"host build {t_host:.2f} s", flush=True)
    host.save("/tmp/h.mz"); dev.save("/tmp/d.mz")
    rc = subprocess.call(["cmp", "/tmp/h.mz", "/tmp/d.mz"])
    print("cmp exit", rc, os.path.getsize("/tmp/h.mz"), flush=True)
    os.unlink("/tmp/h.mz"); os.unlink("/tmp/d.mz")
    assert rc == 0
