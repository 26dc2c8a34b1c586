"""world_size-2 gloo tests (CPU) of the multi-GPU path: batch sharding, the one broadcast of the text
embeddings, latent all-gather, max-over-ranks timing reduction, and shard-vs-global noise identity."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from diffusynth_amd import dist as D
    from diffusynth_amd.sampler import DiffSynthSampler
    r, w, dev = D.init(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    # 1) only rank 0 owns the embeddings; every rank ends up with them after ONE broadcast
    cond = torch.arange(512, dtype=torch.float32) if rank == 0 else None
    unc = -torch.arange(512, dtype=torch.float32) if rank == 0 else None
    c, u = D.broadcast_conditions(cond, unc, dev)
    ok = torch.equal(c, torch.arange(512, dtype=torch.float32)) and torch.equal(u, -torch.arange(512, dtype=torch.float32))
    c2, u2 = D.broadcast_conditions(cond, None, dev)
    ok = ok and u2 is None and torch.equal(c2, c)
    # 2) shard ranges tile the global batch
    lo, hi = D.shard_range(6, rank, world)
    ok = ok and (lo, hi) == (3 * rank, 3 * rank + 3)
    # 3) sharded noise == slice of the global draw (same seed on every rank)
    torch.manual_seed(11)
    s = DiffSynthSampler(1000, mute=True, device="cpu", height=4, max_batchsize=3, channels=2, shard=(rank, world))
    local, _ = s.get_deterministic_noise_tensor(3, 100)
    torch.manual_seed(11)
    g = DiffSynthSampler(1000, mute=True, device="cpu", height=4, max_batchsize=6, channels=2)
    full, _ = g.get_deterministic_noise_tensor(6, 100)
    ok = ok and torch.equal(local, full[lo:hi])
    # 3b) the same with max_batchsize > batch (the reference's default: max_batchsize=16 for any batch <= 16)
    torch.manual_seed(12)
    s5 = DiffSynthSampler(1000, mute=True, device="cpu", height=4, max_batchsize=5, channels=2, shard=(rank, world))
    local5, _ = s5.get_deterministic_noise_tensor(3, 64)
    torch.manual_seed(12)
    g10 = DiffSynthSampler(1000, mute=True, device="cpu", height=4, max_batchsize=10, channels=2)
    full10, _ = g10.get_deterministic_noise_tensor(6, 64)
    ok = ok and torch.equal(local5, full10[lo:hi])
    # 4) all-gather of the per-rank latents restores rank order; timing reduction takes the max
    gathered = D.gather_latents(local)
    ok = ok and torch.equal(gathered, full)
    ok = ok and D.max_over_ranks(1.0 + rank, dev) == float(world)
    D.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_gloo_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_shard_range_rejects_uneven_batches():
    from diffusynth_amd import dist as D
    with pytest.raises(AssertionError):
        D.shard_range(7, 0, 2)
    assert D.shard_range(8, 3, 4) == (6, 8)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts the ranks (torch.distributed.run),
    relays exactly ONE JSON line from rank 0 and returns the children's exit code.  --dry-run = rendezvous + the one
    broadcast + barriers + max-over-ranks only (the product path itself has no CPU mode)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DS_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["value"] is None and out["dry_run"] is True
    assert out["config"]["comm"] == {"backend": "gloo", "ranks": 2}
    # with N > 1 and no --workload the bench runs BASELINE configs[3]'s per-GPU share (64 per GPU, CFG 6, 100-step DDIM) in the parity tier
    assert out["config"]["would_run"] == "config4" and out["config"]["would_run_baseline_config"] == 3 and out["config"]["dtype"] == "bf16x3"
    sys.path.insert(0, root)
    import bench
    assert bench.WORKLOADS["config4"] == (3, 64, 6.0, "ddim", 100, True)
    # bad arguments fail in the parent, before any rank is started
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "nope"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert q.returncode != 0
