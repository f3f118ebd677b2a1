#!/usr/bin/env python3
"""bench.py -- M k-mers/s looked up through the SSHash k2u path on N MI355X (one process per GPU).

A "step" is one pass of the hot path (mz_sshash_k2u_batch_device: canonical encode ->
minimizer -> MPHF -> Elias-Fano bucket -> packed positions -> 2-bit verify -> unitig locate)
over one batch of synthetic queries that is already resident in HBM: 50 % positives drawn
from the index's own unitigs (random orientation), 50 % uniform random negatives, shuffled.
The index is replicated per GPU, the batch is sharded by rank (weak scaling: fixed queries
per GPU), and the only collective is the final all-reduce of {n_hit, n_miss} over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s-human|s-eighth|s-yeast] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import mazu_amd as M  # noqa: E402
from mazu_amd import synth  # noqa: E402

B_ALG = 208.0        # algorithmic bytes per k-mer query on the 50/50 mix (table in DESIGN.md)
B_ALG_OWN = 134.0    # same accounting for this repository's layout
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

DEFAULT_BATCH = {"s-tiny": 200_000, "s-yeast": 10_000_000, "s-txome": 50_000_000, "s-eighth": 50_000_000, "s-human": 125_000_000}


def log(msg):
    print(f"[bench r{os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(us_host, k, w, skew, seed, q_host, n_threads, target_s=8.0):
    """oracle ('port' of sshash.rs) timed on this box's host cores over a bounded sample: a pilot of
    1 M queries sizes the sample so that the timed region lasts about `target_s` seconds"""
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
    touch_bytes = 8.0 * orc.word_touches(q_host[:min(n, 500_000)])[0]  # measured 8-byte words per query, reference layout
    return out, n_hit, n, n / dt / 1e6, t_build, dt, touch_bytes


def bench_reads(args, ix, us, dev, rank, world, D):
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
    torch.cuda.synchronize(dev)
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    D.allreduce_counts(counts)
    torch.cuda.synchronize(dev)
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    n_hit, n_miss, n_skip = counts.cpu().tolist()
    if rank == 0:
        total = n_win * world * args.steps
        res = {"metric": "M k-mers/sec looked up (reads, every window)", "value": round(total / elapsed / 1e6, 2), "unit": "M k-mers/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": f"{args.workload}: {args.n_reads} x {args.read_len} bp reads per GPU per step (50% with 1% substitutions, "
                                      f"random strand, drawn across unitig junctions), SSHash k=31 m={args.w} skew_param={args.skew}",
                          "windows_per_gpu_per_step": n_win, "n_kmers": ix.n_kmers()},
               "hit_fraction": round(n_hit / max(1, n_hit + n_miss), 4), "skipped": n_skip}
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
            assert same, "GPU read results differ from the oracle's streaming walk"
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
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
    ap.add_argument("--neg-frac", type=float, default=0.5, help="fraction of NEG-U negatives in the query mix (the metric is quoted at 0.5)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries in the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-validate", dest="validate", action="store_false",
                    help="skip the correctness gate (K2U::validate_self on the device, every k-mer in both orientations) before timing")
    ap.add_argument("--host-build", action="store_true", help="build the index with the multi-threaded host builder instead of on the GPU")
    ap.add_argument("--mode", default="kmers", choices=["kmers", "reads"],
                    help="kmers: batched point queries (headline); reads: BASELINE config 3, every window of 150-bp reads "
                         "through mz_sshash_k2u_reads_device (.as_streaming() semantics)")
    ap.add_argument("--n-reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--err-frac", type=float, default=0.5, help="reads mode: fraction of reads with 1%% substitutions")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    n_gpus = world
    if M.lib.mz_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the k2u path has no CPU fallback")
    # one process per GPU; MZ_DIST_BACKEND=gloo lets several ranks share one card to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("MZ_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    from mazu_amd import dist as D
    D.init(backend, dev)  # nccl == RCCL on ROCm; no-op for a single process

    k = 31
    batch = args.batch or DEFAULT_BATCH[args.workload]
    # ---- unitig set (same seed on every rank -> identical replica), index build on the host
    t0 = time.time()
    us = synth.make_unitig_set(args.workload, k=k, device=dev)
    words_h = us["words"].cpu().numpy().view(np.uint64)
    starts_h = us["starts"].cpu().numpy().view(np.uint64)
    t_gen = time.time() - t0
    t0 = time.time()
    shared = None
    if args.host_build and world > 1:  # one host build per node: rank 0 builds and lends the flat index file through /dev/shm
        shared = f"/dev/shm/mazu_amd_bench_{os.environ.get('MASTER_PORT', '0')}_{args.workload}.mzsshash"
    if not args.host_build:  # every rank builds its replica on its own GPU (bit-identical arrays; host builder where unsupported)
        uset = M.UnitigSet.from_packed(words_h, us["total_len"], starts_h, k)
        ix = M.SSHash.from_unitig_set_on_device(uset, args.w, args.skew, args.seed, device=dev_index)
        del uset
    else:
        if rank == 0:
            uset = M.UnitigSet.from_packed(words_h, us["total_len"], starts_h, k)
            ix = M.SSHash.from_unitig_set(uset, args.w, args.skew, args.seed)
            del uset
            if shared:
                ix.save(shared)
        if world > 1:
            D.barrier()
            if rank != 0:
                ix = M.SSHash.load(shared)
            D.barrier()
            if rank == 0:
                os.unlink(shared)
    t_build = time.time() - t0
    t0 = time.time()
    ix.to_device(dev_index)
    torch.cuda.synchronize(dev)
    t_upload = time.time() - t0
    log(f"{args.workload}: {us['total_len']} bp, {us['n_unitigs']} unitigs, {ix.n_kmers()} k-mers; gen {t_gen:.1f}s "
        f"build {t_build:.1f}s upload {t_upload:.1f}s; {ix.n_minimizers()} minimizers, {ix.n_minimizer_occs()} occs, "
        f"{ix.n_kmers_in_skew_index()} skew k-mers, {ix.device_bytes() / 1e9:.2f} GB in HBM, "
        f"{ix.num_bits() / ix.n_kmers():.2f} bits/k-mer")
    validate = None
    if args.validate:
        t0 = time.time()
        validate = ix.validate_self_report()
        validate["seconds"] = time.time() - t0
        log(f"validate_self: {validate}")
        assert validate["n_fail"] == 0, validate

    if args.mode == "reads":
        return bench_reads(args, ix, us, dev, rank, world, D)
    # ---- this rank's shard of the query batch, generated in HBM
    q = synth.make_query_mix(ix, us, batch, seed=1000 + rank, device=dev, neg_frac=args.neg_frac, neg_kind=args.neg_kind)
    out = torch.empty(batch * 2, dtype=torch.int64, device=dev)
    counts = torch.zeros(3, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        ix.k2u_batch_device(q, out, counts, stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(dev)
        D.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    counts.zero_()
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
        tf = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
        if os.path.exists(tf) and args.neg_frac == 0.5 and args.neg_kind == "uniform":  # HBM bytes per launch from the committed rocprofv3 --pmc passes (same workload/batch only)
            rec = json.load(open(tf)).get(f"{args.workload}:{batch}")
            traffic = rec["hbm_bytes_per_launch"] if rec else None
        res = {
            "metric": "M k-mers/sec looked up (whole node), GRCh38 k=31, at 1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "M k-mers/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: synthetic GRCh38-shaped unitig set, SSHash k=31 m={args.w} skew_param={args.skew}, "
                                   f"{batch} queries/GPU/step, {100 - round(100 * args.neg_frac)}/{round(100 * args.neg_frac)} positive/negative" + ("" if args.neg_kind == "uniform" else " (NEG-1 negatives)"),
                       "total_len_bp": us["total_len"], "n_unitigs": us["n_unitigs"], "n_kmers": ix.n_kmers(),
                       "queries_per_gpu_per_step": batch, "index_hbm_gb": round(ix.device_bytes() / 1e9, 3),
                       "parallelism": f"index replicated x{n_gpus}, queries sharded, all-reduce(hit counts)"},
            "hit_fraction": round(n_hit / total_q, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "frac_own_layout": round(batch * B_ALG_OWN / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic_gbs": round(traffic / avg_kernel_s / 1e9, 1) if traffic else None,
                         "kernel": {"v1": "k2u_batch_kernel", "tile": "k2u_tile_kernel"}.get(os.environ.get("MZ_K2U_KERNEL", ""), "k2u_wave_kernel"), "kernel_ms_avg": round(avg_kernel_s * 1e3, 4),
                         "algorithmic_bytes_per_kmer": B_ALG},
            "build": {"gen_s": round(t_gen, 2), "where": "host" if args.host_build else "gpu", "build_s": round(t_build, 2), "upload_s": round(t_upload, 2),
                      "bits_per_kmer": round(ix.num_bits() / ix.n_kmers(), 2)},
        }
        if validate is not None:
            res["validate_self"] = validate
        if n_gpus == 1 and not args.no_cpu_baseline:
            cores = M.lib.mz_host_threads()  # hardware threads capped by the cgroup CPU quota of this box
            q_host = q[: (args.cpu_sample or batch)].cpu().numpy().view(np.uint64)
            want, orc_hits, n_sample, cpu_rate, t_ob, t_q, touch_bytes = cpu_baseline((words_h, us["total_len"], starts_h), k, args.w, args.skew,
                                                                       args.seed, q_host, cores)
            got = out[: 2 * n_sample].cpu().numpy().view(M.K2UPOS_DTYPE)
            same = got.tobytes() == want.tobytes()
            res["cpu_baseline"] = {"value": round(cpu_rate, 3), "unit": "M k-mers/s", "cores": cores, "kind": "port",
                                   "sample": f"first {n_sample} queries of the rank-0 batch on the same {args.workload} index, "
                                             f"OpenMP static shards over {cores} threads (oracle build {t_ob:.1f}s, query {t_q:.1f}s)",
                                   "results_identical_to_gpu": same,
                                   "word_touch_bytes_per_kmer": round(touch_bytes, 1)}
            res["gpu_over_cpu"] = round(value / cpu_rate, 1)
            assert same, "GPU results differ from the CPU oracle on the sampled queries"
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
