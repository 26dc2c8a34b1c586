"""Multi-GPU sampling: one process per GPU, batch sharded over ranks, no data-path collective.

Every sample's trajectory depends only on its own noise, the (replicated) weights and the shared
(cond, uncond) text embeddings, so ranks never exchange activations.  The only
communication is one RCCL broadcast of the packed embeddings from rank 0 per request
(``broadcast_conditions``) and an optional all-gather of the final latents.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the launcher's environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).
    Returns (rank, world, device).  backend defaults to nccl (= RCCL on ROCm) when a GPU is visible, else gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if backend is None:
        # DS_DIST_BACKEND / DS_DIST_SHARE_GPU exist only to rehearse the N>1 path on a one-GPU box (gloo, all ranks on cuda:0)
        backend = os.environ.get("DS_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    if os.environ.get("DS_DIST_SHARE_GPU") == "1":
        local = 0
    device = torch.device(f"cuda:{local}") if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of ``total`` samples owned by ``rank`` (total must divide evenly)."""
    assert total % world == 0, f"global batch {total} must be a multiple of the world size {world}"
    per = total // world
    return rank * per, (rank + 1) * per


def broadcast_conditions(cond, uncond, device, label_dim=512, src=0):
    """Rank ``src`` supplies cond (label_dim,) and uncond (label_dim,) (uncond may be None -> zeros + flag);
    every rank returns both on ``device``.  One broadcast of a packed (2*label_dim+1,) fp32 buffer."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    buf = torch.zeros(2 * label_dim + 1, dtype=torch.float32, device=device)
    if rank == src:
        buf[:label_dim] = cond.to(device=device, dtype=torch.float32).reshape(-1)
        if uncond is not None:
            buf[label_dim:2 * label_dim] = uncond.to(device=device, dtype=torch.float32).reshape(-1)
            buf[-1] = 1.0
    if world > 1:
        dist.broadcast(buf, src=src)
    has_uncond = bool(buf[-1].item() == 1.0)
    return buf[:label_dim].clone(), (buf[label_dim:2 * label_dim].clone() if has_uncond else None)


def gather_latents(local):
    """All-gather the per-rank final latents (B_local, C, H, W) into the global batch, rank order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    out = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local.contiguous())
    return torch.cat(out, 0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()
