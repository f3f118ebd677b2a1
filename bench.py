#!/usr/bin/env python3
"""bench.py -- M k-mers/s looked up through the SSHash k2u path on N MI355X (one process per GPU).

A "step" is one pass of the hot path (mz_sshash_k2u_batch_device: canonical encode ->
minimizer -> MPHF -> Elias-Fano bucket -> packed positions -> 2-bit verify -> unitig locate)
over one batch of synthetic queries that is already resident in HBM: 50 % positives drawn
from the index's own unitigs (random orientation), 50 % uniform random negatives, shuffled.
The index is replicated per GPU, the batch is sharded by rank (weak scaling: fixed queries
per GPU), and the only collective is the final all-reduce of {n_hit, n_miss} over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s-human|s-pan|s-eighth|s-yeast] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 outside a torchrun environment starts the N ranks itself (fresh child processes, before this
process has made any GPU call) and relays rank 0's JSON line.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

B_ALG = 208.0        # algorithmic bytes per k-mer query on the 50/50 mix, reference layout: MPHF levels + Elias-Fano offsets + bucket entries + 2-bit verify + boundary rank/select, each distinct word once
B_ALG_OWN = {True: 178.0, False: 122.0}  # same accounting for this repository's layout: with minimizer lines / minimizer records only
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, 8.0 TB/s spec

DEFAULT_BATCH = {"s-tiny": 200_000, "s-yeast": 10_000_000, "s-txome": 50_000_000, "s-eighth": 50_000_000, "s-human": 125_000_000,
                 "s-pan": 125_000_000, "s-human-rep": 125_000_000, "s-eighth-rep": 50_000_000, "s-tiny-rep": 200_000}
SHAPE = {"s-human": "GRCh38-shaped", "s-pan": "pan-genome-shaped (120 M unitigs, 9 Gbp)", "s-eighth": "GRCh38/8-shaped",
         "s-yeast": "yeast-sized", "s-txome": "transcriptome-shaped", "s-tiny": "tiny",
         "s-human-rep": "GRCh38-shaped, repeat families (8 % of the sequence, 300 bp, 5-15 % divergence)",
         "s-eighth-rep": "GRCh38/8-shaped, repeat families", "s-tiny-rep": "tiny, repeat families"}


def log(msg):
    print(f"[bench r{os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("MZ_WORKLOAD", "s-human"))
    ap.add_argument("--batch", type=int, default=0, help="queries per GPU per step")
    ap.add_argument("--w", type=int, default=15)
    ap.add_argument("--skew", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--neg-kind", default="uniform", choices=["uniform", "sub1"],
                    help="negatives: NEG-U uniform words (the headline) or NEG-1 positives with one substituted base")
    ap.add_argument("--neg-frac", type=float, default=0.5, help="fraction of negatives in the query mix (the metric is quoted at 0.5)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries in the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-mixes", action="store_true", help="skip the NEG-1 and positives-only side measurements")
    ap.add_argument("--no-validate", dest="validate", action="store_false",
                    help="skip the correctness gate (K2U::validate_self on the device, every k-mer in both orientations) before timing")
    ap.add_argument("--no-build-check", dest="build_check", action="store_false",
                    help="do not validate the index after the build either (stage-cost experiments with kernel variants whose results are meaningless)")
    ap.add_argument("--lines", default="auto", choices=["auto", "on", "off"],
                    help="minimizer lines: by the size rule, forced, or never built (mz_build_opts.minimizer_lines); off measures the MPHF path alone")
    ap.add_argument("--host-build", action="store_true", help="build the index with the multi-threaded host builder instead of on the GPU")
    ap.add_argument("--mode", default="kmers", choices=["kmers", "reads"],
                    help="kmers: batched point queries (headline); reads: BASELINE config 3, every window of 150-bp reads "
                         "through mz_sshash_k2u_reads_device (.as_streaming() semantics)")
    ap.add_argument("--n-reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--err-frac", type=float, default=0.5, help="reads mode: fraction of reads with 1%% substitutions")
    ap.add_argument("--launch-check", action="store_true",
                    help="only rehearse the launch: every rank joins the process group, the tallies are all-reduced, rank 0 prints a JSON line "
                         "(no index, no GPU work; with MZ_DIST_BACKEND=gloo it runs on a box without GPUs)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`bench.py --gpus N` outside torchrun: start N ranks as fresh child processes (this process has made no GPU call and
    makes none), rank 0 inherits stdout, the others' stdout goes to stderr; exit non-zero if any rank fails."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench launcher] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def query_seed(rank):
    """seed of a rank's query shard (point queries; reads use 2000 + rank): different on every rank, the same from run to run"""
    return 1000 + rank


def launch_check(args):
    """the N>1 plumbing without an index: process group, the path's one collective, max-over-ranks timing"""
    import torch
    from mazu_amd import dist as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = os.environ.get("MZ_DIST_BACKEND", "nccl")
    dev = None
    if backend == "nccl":
        if torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} GPU(s) visible; one process per GPU needs {world}")
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
    D.init(backend, dev)
    warm_collectives(D, dev or "cpu", torch) if dev is not None else (D.allreduce_counts(torch.zeros(3, dtype=torch.int64)), D.barrier())
    counts = torch.tensor([rank + 1, 10 * (rank + 1), 0], dtype=torch.int64, device=dev or "cpu")
    D.allreduce_counts(counts)
    t = D.max_over_ranks(float(rank + 1), dev or "cpu")
    # every rank draws its shard from its own seed (1000 + rank in the real run): the seeds of all ranks, gathered through the same collective
    seeds = torch.zeros(world, dtype=torch.int64, device=dev or "cpu")
    seeds[rank] = query_seed(rank)
    D.allreduce_counts(seeds)
    # the accounting of the real run on a pretend batch: value x ms_per_step == world x batch
    batch, steps, elapsed = 1000, 4, 0.25 * t
    value, ms_per_step = batch * world * steps / elapsed / 1e6, elapsed / steps * 1e3
    D.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "backend": backend, "counts": counts.cpu().tolist(), "max_over_ranks": t,
                          "query_seeds": seeds.cpu().tolist(), "value": value, "ms_per_step": ms_per_step, "batch": batch,
                          "collective": collective_info(backend, world)}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def collective_info(backend, world):
    """what carried the path's one exchange step: read from the process group itself, so that the line says whether RCCL saw N ranks"""
    import torch
    import torch.distributed as dist
    info = {"backend": backend if world > 1 else "none (single process)", "world_size": world, "rccl_version": None}
    if world > 1 and dist.is_initialized():
        info["backend"] = dist.get_backend()
        info["world_size"] = dist.get_world_size()
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:  # torch built without the binding
        pass
    return info


def warm_collectives(D, dev, torch):
    """the path's collectives once, untimed, on scratch tensors: the first all_reduce of a communicator sets up its rings and
    buffers (milliseconds), which must not land inside a timed region of a tenth of a second"""
    scratch = torch.zeros(3, dtype=torch.int64, device=dev)
    D.allreduce_counts(scratch)
    D.max_over_ranks(0.0, dev)
    torch.cuda.synchronize(dev)
    D.barrier()


def kernel_source_hash():
    """hash of the sources every kernel of the hot path is compiled from: a traffic figure measured under the profiler is only
    printed for the kernels it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("mz_device.hip", "mz_common.hpp"):
        with open(os.path.join(ROOT, "mazu_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_record(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of the same command (profiles/traffic_per_launch.json), or None
    when there is none for this workload, or when the kernels have changed since it was taken"""
    tf = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
    if not os.path.exists(tf):
        return None
    rec = json.load(open(tf)).get(key)
    if not rec or rec.get("kernel_source_hash") != kernel_source_hash():
        return None
    return rec


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(us_host, k, w, skew, seed, q_host, n_threads, target_s=8.0):
    """oracle ('port' of sshash.rs) timed on this box's host cores over a bounded sample: a pilot of
    1 M queries sizes the sample so that the timed region lasts about `target_s` seconds; then the single-threaded
    rate (the reference's own bench loop is single-threaded, bin/kphf/main.rs:273-339) on a smaller sample"""
    os.environ["OMP_NUM_THREADS"] = str(n_threads)  # before libgomp initialises inside liboracle.so
    from oracle import oracle as O
    t0 = time.time()
    orc = O.OracleSSHash(us_host[0], us_host[1], us_host[2], k, w, skew, seed, with_dict=False)
    t_build = time.time() - t0
    n_pilot = min(len(q_host), 1_000_000)
    orc.k2u_batch(q_host[:n_pilot], n_threads=n_threads)  # warms threads and caches
    t0 = time.time()
    orc.k2u_batch(q_host[:n_pilot], n_threads=n_threads)
    pilot_rate = n_pilot / max(time.time() - t0, 1e-6)
    n = int(min(len(q_host), max(n_pilot, pilot_rate * target_s)))
    t0 = time.time()
    out, n_hit = orc.k2u_batch(q_host[:n], n_threads=n_threads)
    dt = time.time() - t0
    n1 = int(min(n, max(100_000, pilot_rate / max(n_threads, 1) * 3.0)))  # about 3 s on one thread
    t0 = time.time()
    orc.k2u_batch(q_host[:n1], n_threads=1)
    dt1 = time.time() - t0
    touch_bytes = 8.0 * orc.word_touches(q_host[:min(n, 500_000)])[0]  # measured 8-byte words per query, reference layout
    return dict(out=out, n_hit=n_hit, n=n, rate=n / dt / 1e6, t_build=t_build, dt=dt, touch_bytes=touch_bytes,
                n_single=n1, ns_single=dt1 / n1 * 1e9)


# ------------------------------------------------------------------------------------------------ projection (the step after k2u)
def bench_projection(ix, us, d_recs, n, dev, torch, np, steps=5):
    """mz_u2pos_project_device over the K2UPos records of the headline batch; returns M MappedRefPos records per second"""
    import ctypes as C
    from mazu_amd._capi import check, lib
    n_u = us["n_unitigs"]
    starts = us["starts"]
    # one forward/backward occurrence per unitig on references of 2^30 bases cut from the unitig concatenation
    ref_log = min(30, int(us["total_len"]).bit_length())
    pos, ref = starts[:-1] & ((1 << ref_log) - 1), starts[:-1] >> ref_log
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    fwd = torch.randint(0, 2, (n_u,), generator=g, device=dev, dtype=torch.int64)
    pos_bits, ref_bits = ref_log, max(1, int(us["total_len"] >> ref_log).bit_length())
    width, ref_shift = 1 + pos_bits + ref_bits, pos_bits + 1
    enc = (ref << ref_shift) | (pos << 1) | fwd                          # UnitigOcc::encode_piscem, src/spt_compact.rs:90-96
    bit = torch.arange(n_u, device=dev, dtype=torch.int64) * width
    wi, sh = bit >> 6, bit & 63
    words = torch.zeros((n_u * width + 63) // 64 + 2, dtype=torch.int64, device=dev)
    words.index_add_(0, wi, enc << sh)                                   # disjoint bit fields: addition is bitwise or
    spill = (sh + width) > 64
    words.index_add_(0, wi[spill] + 1, enc[spill] >> (64 - sh[spill]))
    offs = np.arange(n_u + 1, dtype=np.uint64)
    words_h = words.cpu().numpy().view(np.uint64)
    u = C.c_void_p()
    check(lib.mz_u2pos_piscem_new(words_h.ctypes.data, n_u, width, ref_shift, (1 << pos_bits) - 1, offs.ctypes.data, 64, n_u, C.byref(u)))
    try:
        check(lib.mz_u2pos_to_device(u, dev.index))
        stream = torch.cuda.current_stream(dev)
        d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        d_out = torch.empty(2 * n, dtype=torch.int64, device=dev)        # at most one record per query here
        k = us["k"]

        def step():
            check(lib.mz_u2pos_project_device(u, k, d_recs.data_ptr(), n, d_off.data_ptr(), d_out.data_ptr(), n, stream.cuda_stream))
        step()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(steps):
            step()
        b.record(stream)
        torch.cuda.synchronize(dev)
        total = int(d_off[n].item())
        ms = a.elapsed_time(b) / steps
        return {"table": f"PiscemUnitigTable, {n_u} unitigs x 1 occurrence, {width} bits each", "records_per_launch": total, "ms_per_launch": round(ms, 4),
                "M_records_per_s": round(total / ms / 1e3, 1), "M_queries_per_s": round(n / ms / 1e3, 1)}
    finally:
        lib.mz_u2pos_free(u)


# ------------------------------------------------------------------------------------------------ realistic skew (side line)
def bench_realistic_skew(args, batch, dev, synth, torch, workload="s-human-rep", w=None):
    """point-query rates and the layout statistics of a variant of the headline configuration (built, validated and freed here): the
    repeat-family form of the set, or the same set at another minimizer length"""
    t0 = time.time()
    w = w or args.w
    us2 = synth.make_unitig_set(workload, k=31, device=dev)
    ix2, rep, rebuilds = synth.build_index_unique(us2, w, args.skew, args.seed, device=dev.index, on_device=not args.host_build)
    st = ix2.line_stats()
    out = torch.empty(batch * 2, dtype=torch.int64, device=dev)
    cnt = torch.zeros(3, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    rates = {}
    for name, kw in (("neg_u_50_50", {}), ("neg_1_50_50", dict(neg_kind="sub1")), ("positives_only", dict(neg_frac=0.0))):
        q2 = synth.make_query_mix(ix2, us2, batch, seed=3000, device=dev, **kw)
        ix2.k2u_batch_device(q2, out, cnt, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(5):
            ix2.k2u_batch_device(q2, out, cnt, stream.cuda_stream)
        b.record(stream)
        torch.cuda.synchronize(dev)
        rates[name] = round(batch * 5 / (a.elapsed_time(b) / 1e3) / 1e6, 2)
        del q2
    what = f"{workload}: {SHAPE[workload]}, m={w}" + (f"; {us2['rep_substitutions']} bases substituted so that no 31-mer of a copy repeats" if us2.get("rep_substitutions") else "")
    return {"workload": what, "built_on": "gpu" if ix2.built_on_device() else "host",
            "M_kmers_per_s_one_gpu": rates, "validate_self": rep, "n_kmers": ix2.n_kmers(), "index_hbm_gb": round(ix2.device_bytes() / 1e9, 3),
            "lines": {"g": st["g"], "t": st["t"], "entries_per_line": round(st["entries_per_line"], 3), "frac_inline": round(st["frac_inline"], 4),
                      "frac_spilled": round(st["frac_spilled"], 4), "frac_heavy": round(st["frac_heavy"], 4)},
            "seconds": round(time.time() - t0, 1)}


# ------------------------------------------------------------------------------------------------ reads mode
def bench_reads(args, ix, us, dev, rank, world, D, M, synth, np, torch):
    """BASELINE config 3: reads -> every k-mer window, results identical to independent k2u calls"""
    k = us["k"]
    bases, ro = synth.make_reads(us, args.n_reads, args.read_len, seed=2000 + rank, device=dev, err_frac=args.err_frac)
    ko = synth.kmer_offsets(ro, k)
    n_win = int(ko[-1])
    out = torch.empty(n_win * 2, dtype=torch.int64, device=dev)
    counts = torch.zeros(3, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        ix.k2u_reads_device(bases, ro, ko, out, counts, streaming=True, stream=stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    counts.zero_()
    warm_collectives(D, dev, torch)
    torch.cuda.synchronize(dev)
    D.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    D.allreduce_counts(counts)
    torch.cuda.synchronize(dev)
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    kernel_s = float(np.mean([a.elapsed_time(b) for a, b in ev])) / 1e3
    n_hit, n_miss, n_skip = counts.cpu().tolist()
    if rank == 0:
        total = n_win * world * args.steps
        res = {"metric": "M k-mers/sec looked up (reads, every window)", "value": round(total / elapsed / 1e6, 2), "unit": "M k-mers/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": f"{args.workload}: {args.n_reads} x {args.read_len} bp reads per GPU per step (50% with 1% substitutions, "
                                      f"random strand, drawn across unitig junctions), SSHash k=31 m={args.w} skew_param={args.skew}",
                          "windows_per_gpu_per_step": n_win, "n_kmers": ix.n_kmers()},
               "hit_fraction": round(n_hit / max(1, n_hit + n_miss), 4), "skipped": n_skip, "timed_region_s": round(elapsed, 4),
               "collective": collective_info(os.environ.get("MZ_DIST_BACKEND", "nccl"), world)}
        if world == 1 and not args.no_cpu_baseline:  # parity of a sample of reads against the oracle
            os.environ["OMP_NUM_THREADS"] = str(M.lib.mz_host_threads())
            from oracle import oracle as O
            orc = O.OracleSSHash(us["words"].cpu().numpy().view(np.uint64), us["total_len"], us["starts"].cpu().numpy().view(np.uint64),
                                 k, args.w, args.skew, args.seed, with_dict=False)
            n_s = min(args.n_reads, 20000)
            t0 = time.time()
            want, ko_h, _ = orc.k2u_reads(bases[: n_s * args.read_len].cpu().numpy(), ro[: n_s + 1].cpu().numpy().astype(np.uint64), streaming=True)
            dt = time.time() - t0
            got = out[: 2 * int(ko_h[-1])].cpu().numpy().view(M.K2UPOS_DTYPE)
            same = got.tobytes() == want.tobytes()
            res["cpu_baseline"] = {"value": round(len(want) / dt / 1e6, 3), "unit": "M k-mers/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n_s} reads, single thread, StreamingK2U walk (bin/kphf/main.rs:299-312)",
                                   "results_identical_to_gpu": same}
            # algorithmic bytes of the streaming walk in the reference layout: a cold window costs the whole chain (B_ALG), a warm one
            # a quarter byte of sequence and its 16-byte result; f_cold is counted by the oracle on the sample (match 3 = skipped window)
            f_cold = orc.last_reads_cold / max(1, len(want) - int((want["match"] == 3).sum()))
            b_win = B_ALG * f_cold + 17.0 * (1.0 - f_cold)
            achieved = n_win * b_win / kernel_s / 1e9
            # HBM bytes and VALU wave-instructions per launch from the committed rocprofv3 --pmc passes (same workload and kernels only).  `frac`
            # prices the windows at what the REFERENCE's walk would move; `traffic_gbs` is what this kernel moves, and `issue` says what bounds
            # it: VALU wave-instructions per 64 windows, and the fraction of the SIMDs' issue slots they fill (4 cycles each at 2.4 GHz)
            rec = traffic_record(f"reads:{args.workload}:{n_win}") if args.err_frac == 0.5 and not any(ix.get_query_opts().values()) else None
            res["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": rec["hbm_bytes_per_launch"] if rec else None,
                               "traffic_gbs": round(rec["hbm_bytes_per_launch"] / kernel_s / 1e9, 1) if rec else None,
                               "kernel": "k2u_reads_wave_kernel" if args.w in (15, 17, 19, 21) else "k2u_reads_kernel", "kernel_ms_avg": round(kernel_s * 1e3, 4),
                               "algorithmic_bytes_per_window": round(b_win, 1), "f_cold": round(f_cold, 4)}
            if rec and rec.get("valu_wave_insts_per_launch"):
                n_simd, clk = 1024, 2.4e9
                res["roofline"]["issue"] = {"valu_per_64_windows": round(rec["valu_wave_insts_per_launch"] / n_win * 64, 1),
                                            "valu_issue_frac": round(rec["valu_wave_insts_per_launch"] * 4 / (n_simd * clk * kernel_s), 3)}
            assert same, "GPU read results differ from the oracle's streaming walk"
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        D.barrier()  # every rank leaves the group at the same time
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ main
def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # before torch, the library or any GPU call
        sys.exit(self_launch(args, argv))
    sys.path.insert(0, ROOT)
    if args.launch_check:
        return launch_check(args)

    import numpy as np
    import torch
    import mazu_amd as M
    from mazu_amd import synth
    from mazu_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    n_gpus = world
    # one process per GPU; MZ_DIST_BACKEND=gloo lets several ranks share one card to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("MZ_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()  # does not initialise the GPU
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU: the k2u path has no CPU fallback")
    if backend == "nccl" and n_dev < world:
        raise SystemExit(f"bench.py --gpus {world}: only {n_dev} GPU(s) visible; one process per GPU needs {world} "
                         f"(MZ_DIST_BACKEND=gloo rehearses the multi-rank path on fewer cards)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    D.init(backend, dev)  # nccl == RCCL on ROCm; no-op for a single process

    k = 31
    batch = args.batch or DEFAULT_BATCH[args.workload]
    # ---- unitig set (same seed on every rank -> identical replica) and its index, built on this rank's GPU
    t0 = time.time()
    us = synth.make_unitig_set(args.workload, k=k, device=dev)
    t_gen = time.time() - t0
    t0 = time.time()
    # every rank builds its own replica (GPU builder: bit-identical arrays on every rank; --host-build: the multi-threaded
    # host builder); duplicate 31-mers of the random background (a handful at this scale) are broken up and the index rebuilt,
    # so K2U::validate_self holds with no duplicate at all
    ix, validate, rebuilds = synth.build_index_unique(us, args.w, args.skew, args.seed, device=dev_index, on_device=not args.host_build,
                                                      log=log if rank == 0 else None, check=args.build_check,
                                                      minimizer_lines={"auto": 0, "on": 1, "off": -1}[args.lines])
    torch.cuda.synchronize(dev)
    t_build = time.time() - t0
    words_h = us["words"].cpu().numpy().view(np.uint64)
    starts_h = us["starts"].cpu().numpy().view(np.uint64)
    log(f"{args.workload}: {us['total_len']} bp, {us['n_unitigs']} unitigs, {ix.n_kmers()} k-mers; gen {t_gen:.1f}s "
        f"build+validate {t_build:.1f}s ({rebuilds} rebuild(s) for duplicates); {ix.n_minimizers()} minimizers, {ix.n_minimizer_occs()} occs, "
        f"{ix.n_kmers_in_skew_index()} skew k-mers, {ix.device_bytes() / 1e9:.2f} GB in HBM, "
        f"{ix.num_bits() / ix.n_kmers():.2f} bits/k-mer")
    if args.validate:  # the correctness gate: every k-mer of every unitig in both orientations, on the device
        t0 = time.time()
        validate = ix.validate_self_report()
        validate["seconds"] = round(time.time() - t0, 3)
        assert validate["n_fail"] == 0 and validate["n_duplicate"] == 0 and validate["n_kmers"] == ix.n_kmers(), validate

    if args.mode == "reads":
        return bench_reads(args, ix, us, dev, rank, world, D, M, synth, np, torch)
    # ---- this rank's shard of the query batch, generated in HBM
    q = synth.make_query_mix(ix, us, batch, seed=query_seed(rank), device=dev, neg_frac=args.neg_frac, neg_kind=args.neg_kind)
    out = torch.empty(batch * 2, dtype=torch.int64, device=dev)
    counts = torch.zeros(3, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step(qq=q):
        ix.k2u_batch_device(qq, out, counts, stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(dev)
        D.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    counts.zero_()
    warm_collectives(D, dev, torch)  # the communicator's first all_reduce happens here, not inside the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    D.allreduce_counts(counts)  # the path's one exchange step: final hit-count all-reduce over RCCL/xGMI
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    n_hit, n_miss, _ = counts.cpu().tolist()
    total_q = batch * n_gpus * args.steps
    assert n_hit + n_miss == total_q, (n_hit, n_miss, total_q)

    if rank == 0:
        value = total_q / elapsed / 1e6
        avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
        achieved = batch * B_ALG / avg_kernel_s / 1e9
        traffic = None
        if args.neg_frac == 0.5 and args.neg_kind == "uniform" and args.w == 15 and not ix.get_query_opts()["point_kernel"]:  # (same workload, batch and kernel only)
            rec = traffic_record(f"{args.workload}:{batch}")
            traffic = rec["hbm_bytes_per_launch"] if rec else None
        res = {
            "metric": "M k-mers/sec looked up (whole node), GRCh38 k=31, at 1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "M k-mers/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: synthetic {SHAPE.get(args.workload, '')} unitig set, SSHash k=31 m={args.w} skew_param={args.skew}, "
                                   f"{batch} queries/GPU/step, {100 - round(100 * args.neg_frac)}/{round(100 * args.neg_frac)} positive/negative" + ("" if args.neg_kind == "uniform" else " (NEG-1 negatives)"),
                       "total_len_bp": us["total_len"], "n_unitigs": us["n_unitigs"], "n_kmers": ix.n_kmers(),
                       "queries_per_gpu_per_step": batch, "index_hbm_gb": round(ix.device_bytes() / 1e9, 3),
                       "parallelism": f"index replicated x{n_gpus}, queries sharded, all-reduce(hit counts)"},
            "hit_fraction": round(n_hit / total_q, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "frac_own_layout": round(batch * B_ALG_OWN[bool(ix.has_minimizer_lines())] / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic_gbs": round(traffic / avg_kernel_s / 1e9, 1) if traffic else None,
                         "kernel": {3: "k2u_batch_kernel", 2: "k2u_tile_kernel"}.get(ix.get_query_opts()["point_kernel"], "k2u_wave_kernel"), "kernel_ms_avg": round(avg_kernel_s * 1e3, 4),
                         "algorithmic_bytes_per_kmer": B_ALG},
            "timed_region_s": round(elapsed, 4), "collective": collective_info(backend, n_gpus),
            "build": {"gen_s": round(t_gen, 2), "where": "gpu" if ix.built_on_device() else "host", "build_and_validate_s": round(t_build, 2),
                      "rebuilds_for_duplicates": rebuilds, "bits_per_kmer": round(ix.num_bits() / ix.n_kmers(), 2),
                      "lines": {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in ix.line_stats().items()
                                if kk in ("g", "t", "entries_per_line", "frac_inline", "frac_spilled", "frac_heavy")}},
        }
        if validate is not None:
            res["validate_self"] = validate
        if n_gpus == 1 and not args.no_extra_mixes and args.neg_frac == 0.5 and args.neg_kind == "uniform":  # (side measurements on one GPU only: no rank waits for another's)
            # the realistic mixes beside the headline: NEG-1 negatives (a positive with one substituted base: shares minimizer and
            # flanks, so it reaches the k-mer verification) and positives only; same batch size, a few steps each
            mixes = {"neg_u_50_50": round(value, 2)}
            for name, kw in (("neg_1_50_50", dict(neg_frac=0.5, neg_kind="sub1")), ("positives_only", dict(neg_frac=0.0, neg_kind="uniform"))):
                q2 = synth.make_query_mix(ix, us, batch, seed=3000 + rank, device=dev, **kw)
                step(q2)
                torch.cuda.synchronize(dev)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n_extra = max(3, min(args.steps, 10))
                a.record(stream)
                for _ in range(n_extra):
                    step(q2)
                b.record(stream)
                torch.cuda.synchronize(dev)
                mixes[name] = round(batch * n_extra / (a.elapsed_time(b) / 1e3) / 1e6, 2)
                del q2
            res["mixes_M_kmers_per_s_one_gpu"] = mixes
            step()  # `out` holds the headline batch's results again for the parity check below
            torch.cuda.synchronize(dev)
        if n_gpus == 1 and not args.no_extra_mixes and args.workload == "s-human":
            # a side line, never the headline: the same shape with the repeat model of the configuration's description (repeat families with
            # per-copy divergence: bucket sizes form a continuum) instead of planted 16-mers -- does the layout hold up there?
            res["realistic_skew"] = bench_realistic_skew(args, batch, dev, synth, torch)
            if args.w == 15:
                # a second side line: the same set at minimizer length 19 (minimizer words beyond 32 bits: BBHash levels + hash-addressed lines, built on the GPU)
                res["minimizer_19"] = bench_realistic_skew(args, batch, dev, synth, torch, workload="s-human", w=19)
        if n_gpus == 1 and not args.no_extra_mixes:
            # the step after k2u (get_ref_pos: PiscemUnitigTable occurrences -> MappedRefPos, src/index.rs:156-216) on the headline batch's
            # records: a synthetic tiling in which every unitig occurs once, occurrences packed into the minimal number of bits
            res["projection"] = bench_projection(ix, us, out, batch, dev, torch, np)
        if n_gpus == 1 and not args.no_cpu_baseline:
            cores = M.lib.mz_host_threads()  # hardware threads capped by the cgroup CPU quota of this box
            q_host = q[: (args.cpu_sample or batch)].cpu().numpy().view(np.uint64)
            cb = cpu_baseline((words_h, us["total_len"], starts_h), k, args.w, args.skew, args.seed, q_host, cores)
            got = out[: 2 * cb["n"]].cpu().numpy().view(M.K2UPOS_DTYPE)
            same = got.tobytes() == cb["out"].tobytes()
            res["cpu_baseline"] = {"value": round(cb["rate"], 3), "unit": "M k-mers/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
                                   "sample": f"first {cb['n']} queries of the rank-0 batch on the same {args.workload} index, "
                                             f"OpenMP static shards over {cores} threads (oracle build {cb['t_build']:.1f}s, query {cb['dt']:.1f}s)",
                                   "single_thread_ns_per_kmer": round(cb["ns_single"], 1),
                                   "single_thread_sample": f"first {cb['n_single']} queries, one thread (the reference's bench loop, bin/kphf/main.rs:273-339, is single-threaded)",
                                   "results_identical_to_gpu": same,
                                   "word_touch_bytes_per_kmer": round(cb["touch_bytes"], 1)}
            assert same, "GPU results differ from the CPU oracle on the sampled queries"
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        D.barrier()  # every rank leaves the group at the same time
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
