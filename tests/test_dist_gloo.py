"""N>1 path on CPU: two gloo ranks each take their contiguous shard of one query batch (and of one
read set), tally hits on their replica, and the single collective of the path -- the all-reduce of
the tallies -- must reproduce the single-process totals.  The per-rank lookups are played by the
oracle here (no GPU in this container); on GPUs the same plumbing wraps mz_sshash_k2u_batch_device
(bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, H.ROOT)
    sys.path.insert(0, os.path.join(H.ROOT, "tests"))
    from mazu_amd import dist as D
    from oracle import oracle as O
    r, w = D.init("gloo")
    assert (r, w) == (rank, world)
    k, seqs = H.load_cf_seqs(os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7"))
    words, total, starts = H.pack_seqs(seqs)
    orc = O.OracleSSHash(words, total, starts, k, 15, 16, with_dict=False)  # replica on every rank
    q = np.load(os.path.join(tmp, "q.npy"))
    lo, hi = D.shard_bounds(len(q), rank, world)
    out, n_hit = orc.k2u_batch(q[lo:hi])
    counts = torch.tensor([n_hit, (hi - lo) - n_hit, 0], dtype=torch.int64)
    D.allreduce_counts(counts)
    t = D.max_over_ranks(1.0 + rank)
    D.barrier()
    np.save(os.path.join(tmp, f"out{rank}.npy"), out)
    if rank == 0:
        np.save(os.path.join(tmp, "counts.npy"), counts.numpy())
        assert t == float(world)
    import torch.distributed as dist
    dist.destroy_process_group()


def test_sharded_counts_allreduce_world2(tmp_path):
    from mazu_amd import dist as D
    from oracle import oracle as O
    assert [D.shard_bounds(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [D.shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    k, seqs = H.load_cf_seqs(os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7"))
    words, total, starts = H.pack_seqs(seqs)
    q = H.query_mix(words, starts, k, 60001, seed=77)
    np.save(tmp_path / "q.npy", q)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want, n_hit = O.OracleSSHash(words, total, starts, k, 15, 16, with_dict=False).k2u_batch(q)
    counts = np.load(tmp_path / "counts.npy")
    assert counts.tolist() == [n_hit, len(q) - n_hit, 0]
    got = np.concatenate([np.load(tmp_path / "out0.npy"), np.load(tmp_path / "out1.npy")])
    assert got.tobytes() == want.tobytes()  # shards stay in order: concatenation == unsharded result


def _run_bench(extra_env, *argv):
    import subprocess
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_bench_gpus_n_launches_n_ranks_itself():
    """`bench.py --gpus 2` outside torchrun starts two ranks (fresh processes), they meet in one process group, the
    tallies are all-reduced, and the parent's stdout is rank 0's single JSON line"""
    import json
    r = _run_bench({"MZ_DIST_BACKEND": "gloo"}, "--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]  # gloo itself chats on stdout ("[Gloo] Rank 0 is connected ...")
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["counts"] == [3, 30, 0] and res["max_over_ranks"] == 2.0


@pytest.mark.parametrize("world", [4, 8])
def test_bench_launch_check_at_world_4_and_8(world):
    """the launch of the 4- and 8-GPU runs rehearsed on the CPU (gloo): every rank joins, the warm-up collectives and the path's
    all-reduce run, every rank draws from its own seed, and the line's accounting holds: value x ms_per_step == world x batch"""
    import json
    r = _run_bench({"MZ_DIST_BACKEND": "gloo"}, "--gpus", str(world), "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    tri = world * (world + 1) // 2
    assert res["n_gpus"] == world and res["counts"] == [tri, 10 * tri, 0] and res["max_over_ranks"] == float(world)
    assert res["query_seeds"] == [1000 + r_ for r_ in range(world)] and len(set(res["query_seeds"])) == world
    assert abs(res["value"] * 1e6 * res["ms_per_step"] / 1e3 - world * res["batch"]) < 1e-6 * world * res["batch"]
    assert res["collective"]["backend"] == "gloo" and res["collective"]["world_size"] == world


def test_bench_gpus_n_fails_loudly_without_n_gpus():
    """with the RCCL backend and fewer than N visible GPUs the launcher must exit non-zero with a clear message,
    never fall back to one rank silently"""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has 2+ GPUs")
    r = _run_bench({"MZ_DIST_BACKEND": "nccl"}, "--gpus", "2", "--launch-check")
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["point", "reads"])
def test_bench_two_ranks_on_one_card(mode):
    """the whole N = 2 bench path on the GPU -- two self-launched ranks, a replica built by each, sharded batches, the final
    all-reduce of the tallies, max-over-ranks timing -- with gloo standing in for RCCL so that both ranks can share one card"""
    import json
    argv = ["--gpus", "2", "--workload", "s-yeast", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra-mixes"]
    if mode == "reads":
        argv += ["--mode", "reads", "--n-reads", "20000"]
    r = _run_bench({"MZ_DIST_BACKEND": "gloo"}, *argv)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    if mode == "point":
        per_gpu = res["config"]["queries_per_gpu_per_step"]
        assert abs(res["value"] * 1e6 * res["ms_per_step"] / 1e3 - 2 * per_gpu) < 0.01 * per_gpu  # both ranks' queries are counted
        assert 0.49 < res["hit_fraction"] < 0.51  # the all-reduced tallies cover both shards (the launcher asserts hit + miss == all queries)
    else:
        assert res["config"]["windows_per_gpu_per_step"] == 20000 * 120 and res["hit_fraction"] > 0.5


@pytest.mark.gpu
def test_rccl_group_of_one_rank_reduces_the_tallies():
    """the backend the multi-GPU runs use ("nccl" = RCCL, bound to the rank's device) initialises on this box and carries the path's
    collectives -- sum of the tallies, max of the times, barrier -- on device tensors; with one GPU the group has one rank"""
    import subprocess
    code = f"""
import os, sys, torch
sys.path.insert(0, {H.ROOT!r})
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="{_free_port()}")
from mazu_amd import dist as D
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
assert D.init("nccl", dev, force=True) == (0, 1) and dist.get_backend() == "nccl"
c = torch.tensor([5, 7, 0], dtype=torch.int64, device=dev)
D._reduce(c, dist.ReduceOp.SUM)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
D._reduce(t, dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize(dev)
assert c.tolist() == [5, 7, 0] and t.item() == 1.5
dist.destroy_process_group()
print("rccl ok", flush=True)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-2000:]
