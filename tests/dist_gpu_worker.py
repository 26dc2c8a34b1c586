"""Worker of tests/test_hip_fullsize.py::test_two_rank_rehearsal_on_one_gpu (spawned BEFORE it touches the GPU).

Two fresh processes share cuda:0 (DS_DIST_SHARE_GPU=1) and talk over gloo (DS_DIST_BACKEND=gloo): the same code path
as `bench.py --gpus 2` / a 2-GPU serving job — D.init, ONE broadcast of the text embeddings, a sampler with
shard=(rank, world), gather of the final latents — with the collectives on gloo because one card cannot host an
RCCL ring of two ranks.  Rank 0 also runs the unsharded job and checks the gathered shards against it bit for bit."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          DS_DIST_BACKEND="gloo", DS_DIST_SHARE_GPU="1")
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import numpy as np
        import torch

        import bench
        from diffusynth_amd import dist as D
        from diffusynth_amd.sampler import DiffSynthSampler
        from diffusynth_amd.synth import synth_input, synth_state_dict
        from diffusynth_amd.unet import PRODUCTION_CONFIG, ConditionedUnet

        r, w, dev = D.init()
        assert (r, w) == (rank, world) and dev.type == "cuda" and dev.index == 0
        with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
            spec = [(k, tuple(s)) for k, s in json.load(f)["unet_production"]]
        net = ConditionedUnet(**PRODUCTION_CONFIG)
        net.load_state_dict(synth_state_dict(spec))
        net.to(dev)                                    # fp32 parity tier: per-sample results are batch-invariant
        # text embeddings exist on rank 0 only
        c0 = synth_input("rehearsal_cond", (512,)) if rank == 0 else None
        u0 = synth_input("rehearsal_uncond", (512,)) if rank == 0 else None
        cond, uncond = D.broadcast_conditions(c0, u0, dev)
        Bl, H, W, K = 2, 32, 64, 3
        lo, hi = D.shard_range(Bl * world, rank, world)

        def run(B, shard, mb):
            s = DiffSynthSampler(1000, mute=True, device=dev, height=H, max_batchsize=mb, noise_device="cpu", shard=shard)
            s.respace(list(np.linspace(0, 999, K, dtype=np.int32)))
            s.activate_classifier_free_guidance(3.0, uncond)
            imgs, _ = s.sample(net, (B, 4, H, W), return_tensor=True, condition=cond.unsqueeze(0).repeat(B, 1), sampler="ddpm", seed=21)
            return imgs[-1]

        local = run(Bl, (rank, world), 3)                # max_batchsize 3 > local batch 2 on purpose
        gathered = D.gather_latents(local.cpu())         # gloo all_gather has no CUDA path: latents gathered on the host
        ok, detail = True, ""
        if rank == 0:
            full = run(Bl * world, None, 3 * world).cpu()
            ok = torch.equal(gathered, full)
            detail = f"max|d|={float((gathered - full).abs().max()):.3e}"
        assert torch.equal(gathered[lo:hi], local.cpu())
        # the headline tier sharded: a bf16x3 sample's bits depend on the batch it travels in (split-K and attention segment counts follow
        # the batch) — the gathered shards must agree with the unsharded bf16x3 run to the rounding of fp32 partial sums
        net.set_compute_dtype("bf16x3")
        local3 = run(Bl, (rank, world), 3)
        gathered3 = D.gather_latents(local3.cpu())
        if rank == 0:
            full3 = run(Bl * world, None, 3 * world).cpu()
            d3 = float((gathered3 - full3).abs().max() / full3.abs().max())
            d3r = float((gathered3 - full3).norm() / full3.norm())
            ok = ok and d3 < 5e-5 and d3r < 5e-5          # (measured 1.7e-5 / 2.0e-5 after three CFG steps)
            detail += f"; bf16x3 shard vs unsharded: max-norm rel {d3:.2e}, rms rel {d3r:.2e}"
        # the bench's own sharded path (Philox noise) in the HEADLINE tier, three steps: ranks must produce different, finite samples
        el, _ = bench.run_sample(net, dev, rank, world, 2, 6.0, "ddpm", True, cond, uncond, H, W, 3, 1, False)
        t = D.max_over_ranks(el, dev)
        ok = ok and t >= el
        D.barrier()
        q.put((rank, bool(ok), detail))
        torch.distributed.destroy_process_group()
    except Exception as e:                               # surface the failure in the parent instead of a silent timeout
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:]))
        raise
