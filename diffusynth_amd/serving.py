"""Mixed-width note batches: the MIDI arranger and the text2sound tab ask for notes of different
durations, i.e. latents of different widths (webUI/natural_language_guided_4/track_maker.py:245, text2sound.py:84:
``width = int(256 * ((duration + 1) / 4) / 4)``, any integer in [20, 144]).

The reference serves them one ``sample()`` call per note.  Here one call takes the whole request list:
requests of equal width are stacked into one batch (one plan of the engine, which keeps all plans in one bounded arena —
engine.py:_cached_plan), every bucket runs the ordinary DiffSynthSampler loop, and the results come back in request order.
No padding to a common width is involved: a note's convolution borders, GroupNorm counts and attention length are those
of its own width, so each request's result is what its own single-sample call would have produced (bit for bit in the
fp32 tier with the deterministic DDIM sampler, where no per-step noise is drawn)."""
import numpy as np
import torch

from .sampler import DiffSynthSampler


@torch.no_grad()
def sample_mixed_widths(model, requests, steps, *, timesteps=1000, height=128, channels=4, sampler="ddim", cfg_scale=1.0,
                        unconditional_condition=None, device="cuda", noise_device=None, return_trajectory=False):
    """requests: list of dicts ``{"width": int, "condition": (label_dim,) tensor or None, "seed": int}``.
    Returns a list (request order) of final latents (4, height, width) — or of trajectories when asked.

    Each request's initial noise is drawn like a batch-1 reference call with its seed would draw it
    (``torch.manual_seed(seed)``; ``randn((1, C, H, train_width))``); requests of one width then share the loop.

    Only the deterministic sampler is served this way: with ``"ddpm"`` the per-step noise of a bucket would come from ONE
    generator stream, so a request's result would depend on which other requests share its width and on their order (the
    reference serves one call per note, so that case has no reference behaviour to match)."""
    if sampler != "ddim":
        raise NotImplementedError("sample_mixed_widths serves the deterministic 'ddim' sampler only: per-step noise of a shared "
                                  "bucket would make a request's result depend on its bucket mates (got %r)" % (sampler,))
    if cfg_scale != 1.0 and unconditional_condition is None:
        raise ValueError("cfg_scale != 1 needs an unconditional_condition (the negative-prompt embedding)")
    buckets = {}
    for i, r in enumerate(requests):
        buckets.setdefault((int(r["width"]), r.get("condition") is None), []).append(i)
    out = [None] * len(requests)
    for (width, nocond), idxs in buckets.items():
        B = len(idxs)
        s = DiffSynthSampler(timesteps, mute=True, device=device, height=height, max_batchsize=B, channels=channels, noise_device=noise_device)
        s.respace(list(np.linspace(0, timesteps - 1, steps, dtype=np.int32)))
        if cfg_scale != 1.0:
            s.activate_classifier_free_guidance(cfg_scale, unconditional_condition)
        noises = []
        for i in idxs:
            one = DiffSynthSampler(timesteps, mute=True, device=device, height=height, max_batchsize=1, channels=channels, noise_device=noise_device)
            one._seed(int(requests[i]["seed"]))
            noises.append(one._randn((1, channels, height, one.train_width), 1))
        ref_noise = torch.cat(noises, 0)
        cond = None if nocond else torch.stack([requests[i]["condition"].to(device).float() for i in idxs])
        imgs, _ = s.sample(model, (B, channels, height, width), return_tensor=True, condition=cond, sampler=sampler, initial_noise=ref_noise)
        for k, i in enumerate(idxs):
            out[i] = [im[k] for im in imgs] if return_trajectory else imgs[-1][k]
    return out
