"""Runs ON the GPU box: a repeat-family preset end to end -- generation time, substitutions made by the de-duplication, index
layout statistics (mz_line_stats), point rates on the three mixes and the reads rate.   python tools/rep_probe.py s-eighth-rep"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import mazu_amd as M  # noqa: E402
from mazu_amd import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "s-eighth-rep"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
dev = torch.device("cuda", 0)
t = time.time()
us = synth.make_unitig_set(name, device=dev)
torch.cuda.synchronize()
t_gen = time.time() - t
t = time.time()
ix, rep, rounds = synth.build_index_unique(us, 15, 32, device=0, on_device=True, log=lambda m: print(m, flush=True))
t_build = time.time() - t
st = ix.line_stats()
res = {"workload": name, "gen_s": round(t_gen, 1), "rep_substitutions": us.get("rep_substitutions"), "build_validate_s": round(t_build, 1), "rebuilds": rounds,
       "n_kmers": ix.n_kmers(), "n_minimizers": ix.n_minimizers(), "entries": ix.n_minimizer_occs(), "skew_kmers": ix.n_kmers_in_skew_index(),
       "hbm_gb": round(ix.device_bytes() / 1e9, 2), "bits_per_kmer": round(ix.num_bits() / ix.n_kmers(), 1),
       "lines": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}}
out = torch.empty(batch * 2, dtype=torch.int64, device=dev)
cnt = torch.zeros(3, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream(dev)
for mix, kw in (("neg_u_50_50", {}), ("neg_1_50_50", dict(neg_kind="sub1")), ("positives_only", dict(neg_frac=0.0))):
    q = synth.make_query_mix(ix, us, batch, seed=1000, device=dev, **kw)
    ix.k2u_batch_device(q, out, cnt, stream.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(5):
        ix.k2u_batch_device(q, out, cnt, stream.cuda_stream)
    b.record(stream)
    torch.cuda.synchronize()
    res[mix + "_M_per_s"] = round(batch * 5 / (a.elapsed_time(b) / 1e3) / 1e6, 1)
    del q
print(json.dumps(res), flush=True)
